import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from trajectorycrafter_amd import ops
from tools.microbench import timeit, report
BF=torch.bfloat16
g=torch.Generator(device='cuda').manual_seed(0)
M,K,N=35552,3072,12288
x=torch.randn(M,K,device='cuda',dtype=BF,generator=g); w=torch.randn(N,K,device='cuda',dtype=BF,generator=g)*0.02; b=torch.randn(N,device='cuda',dtype=BF,generator=g)
ms=timeit(lambda: ops.bias_gelu_tanh_(F.linear(x,w),b)); report("linear + tcx bias_gelu", ms, flops=2.0*M*N*K)
try:
    ms=timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True)); report("_addmm_activation(gelu)", ms, flops=2.0*M*N*K)
    y1=torch._addmm_activation(b, x, w.t(), use_gelu=True).float(); y2=F.gelu((F.linear(x.float(),w.float(),b.float())),approximate='tanh')
    print("max err vs fp32 tanh-gelu", float((y1-y2).abs().max()), "mean", float((y1-y2).abs().mean()))
    y3=ops.bias_gelu_tanh_(F.linear(x,w),b).float(); print("tcx path err", float((y3-y2).abs().max()), float((y3-y2).abs().mean()))
except Exception as e:
    print("addmm_activation failed:", e)
ms=timeit(lambda: F.linear(x,w,b)); report("linear+bias only", ms, flops=2.0*M*N*K)
ms=timeit(lambda: F.linear(x,w)); report("linear no bias", ms, flops=2.0*M*N*K)
