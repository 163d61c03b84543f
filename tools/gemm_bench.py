"""tcx_gemm_bf16 vs the library GEMM (hipBLASLt through F.linear) on the transformer's shapes.
Usage: python tools/gemm_bench.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from trajectorycrafter_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = 2 * 17776
g = torch.Generator().manual_seed(0)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, N, K, epi in (("qkv", 9216, 3072, 0), ("attn_out+gate", 3072, 3072, 2), ("ff1+gelu", 12288, 3072, 1),
                        ("ff2+gate", 3072, 12288, 2), ("cross_q", 2048, 3072, 0), ("cross_out", 3072, 2048, 0)):
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).to(torch.bfloat16).cuda()
    h = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    gate = torch.randn(2, 2 * N, generator=g).to(torch.bfloat16).cuda()
    fl = 2.0 * M * N * K
    if epi == 0:
        lib = lambda: F.linear(x, w, b)
        mine = lambda: ops.gemm_bf16(x, w, b)
    elif epi == 1:
        lib = lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True)
        mine = lambda: ops.gemm_bf16(x, w, b, epilogue=1)
    else:
        def lib():
            y = F.linear(x, w, b)
            ops.gated_residual_(h.view(2, M // 2, N), y.view(2, M // 2, N), gate[:, :N], gate[:, N:], 226)
        mine = lambda: ops.gemm_bf16(x, w, b, epilogue=2, res=h, gate_v=gate[:, :N], gate_t=gate[:, N:], text_len=226, out=h)
    tl, tm = timeit(lib), timeit(mine)
    print(f"{name:14s} M={M} N={N:5d} K={K:5d}: library {tl:7.3f} ms ({fl / tl / 1e9:6.0f} TF)   tcx_gemm {tm:7.3f} ms ({fl / tm / 1e9:6.0f} TF)", flush=True)
    del x, w, b, h
