"""What a plain device-to-device copy of the LN + modulate working set reaches on this box (the read + write ceiling the HBM-bound
row kernels are measured against): torch elementwise copy and hipMemcpyDtoD of 218 MB bf16, beside tcx_layernorm_modulate and
tcx_qk_layernorm_rope on the product shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from trajectorycrafter_amd import ops
dev = torch.device("cuda:0")
B, S, C = 2, 17776, 3072
x = torch.randn(B, S, C, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
nbytes = 2 * x.numel() * 2
us = t(lambda: y.copy_(x)); print(f"torch copy_ (elementwise kernel / memcpy): {us:.1f} us = {nbytes / us / 1e6:.2f} TB/s")
us = t(lambda: torch.add(x, 0, out=y)); print(f"torch add(x, 0, out=y) (vectorised elementwise): {us:.1f} us = {nbytes / us / 1e6:.2f} TB/s")
g = torch.randn(C, device=dev, dtype=torch.bfloat16); b = torch.randn(C, device=dev, dtype=torch.bfloat16)
mod = torch.randn(B, 4 * C, device=dev, dtype=torch.bfloat16)
sh_v, sc_v, sh_t, sc_t = mod.chunk(4, dim=1)
us = t(lambda: ops.layernorm_modulate(x, g, b, 1e-5, sh_v, sc_v, sh_t, sc_t, text_len=226, out=y)); print(f"tcx_layernorm_modulate [2,17776,3072]: {us:.1f} us = {nbytes / us / 1e6:.2f} TB/s")
H, D = 48, 64
qkv = torch.randn(B, S, 3 * H * D, device=dev, dtype=torch.bfloat16)
q, k, v = (t_.view(B, S, H, D) for t_ in qkv.chunk(3, -1))
gq = torch.ones(D, device=dev, dtype=torch.bfloat16); bq = torch.zeros(D, device=dev, dtype=torch.bfloat16)
cos = torch.rand(S - 226, D, device=dev); sin = torch.rand(S - 226, D, device=dev)
us = t(lambda: ops.qk_layernorm_rope(q, k, gq, bq, gq, bq, cos, sin, 226, 1e-6, q_scale=0.18, want_k_sqmax=True)); print(f"tcx_qk_layernorm_rope q,k [2,17776,48,64]: {us:.1f} us = {2 * nbytes / us / 1e6:.2f} TB/s")
