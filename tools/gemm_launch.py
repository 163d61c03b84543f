"""The transformer's four big GEMM launches (tcx_gemm_bf16 only, M = 2 x 17776 rows, product epilogues), back to back, for counter passes
(tools/pmc_gemm_sq.sh).  usage: python tools/gemm_launch.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectorycrafter_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M, BF = 2 * 17776, torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", dtype=BF, generator=g)
cases = []
for name, N, K, epi in (("qkv", 9216, 3072, 0), ("out+gate", 3072, 3072, 2), ("ff1+gelu", 12288, 3072, 1), ("ff2+gate", 3072, 12288, 2)):
    x, w, b = rn(M, K), rn(N, K) * K ** -0.5, rn(N)
    if epi == 2:
        res, gate = rn(2, 17776, N), rn(2, 2 * N)
        cases.append((name, lambda x=x, w=w, b=b, res=res, gate=gate, N=N: ops.gemm_bf16(x.view(2, 17776, -1), w, b, epilogue=2, res=res, gate_v=gate[:, :N], gate_t=gate[:, N:], text_len=226, out=res)))
    else:
        cases.append((name, lambda x=x, w=w, b=b, epi=epi: ops.gemm_bf16(x, w, b, epilogue=epi)))
torch.cuda.synchronize()
for _ in range(iters):
    for name, fn in cases:
        fn()
torch.cuda.synchronize()
print("done", iters, [c[0] for c in cases])
