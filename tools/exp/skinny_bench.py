import sys, torch
sys.path.insert(0, "/root/repo")
from trajectorycrafter_amd import ops
BF=torch.bfloat16
g=torch.Generator().manual_seed(0)
for M,N,K in ((2,18432,512),(2,6144,512),(2,512,3072),(2,512,512)):
    x=torch.randn(M,K,generator=g).to(BF).cuda(); w=(torch.randn(N,K,generator=g)/K**0.5).to(BF).cuda(); b=torch.randn(N,generator=g).to(BF).cuda()
    x9=torch.randn(9,K,generator=g).to(BF).cuda()
    def t(fn, it=200):
        for _ in range(10): fn()
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
    print(f"M={M} N={N} K={K}: skinny {t(lambda: ops.gemm_bf16(x,w,b)):.1f} us   MFMA tile kernel (M=9) {t(lambda: ops.gemm_bf16(x9,w,b)):.1f} us   weights {N*K*2/1e6:.1f} MB")
