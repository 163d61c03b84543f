"""The two attention launches exactly as the product issues them, back to back, for counter passes (tools/pmc_attn_sq.sh):
  self  : qk_layernorm_rope(q pre-scaled, k_sqmax) -> attn_fwd(log2 scores, bound PROVEN, tail split) on the fused-QKV layout
          [2, 17776, 48, 64] (views of one [2, 17776, 9216] projection output, row stride 18 KB)            -> attn_fwd_kernel<64,...,true>
  cross : scale_sqmax(k) -> attn_fwd(log2 scores, bound TESTED per workgroup) at q [2,17550,16,128] x kv [2,4050,16,128]
                                                                                                             -> attn_fwd_kernel<128,...>
usage: python tools/attn_launch.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectorycrafter_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
BF, LOG2E = torch.bfloat16, 1.4426950408889634
g = torch.Generator(device="cuda").manual_seed(0)
B, S, H, D = 2, 17776, 48, 64
qkv = torch.randn(B, S, 3 * H * D, device="cuda", dtype=BF, generator=g)
q, k, v = (t.view(B, S, H, D) for t in qkv.chunk(3, -1))
gam = torch.ones(D, device="cuda", dtype=BF)
bet = torch.zeros(D, device="cuda", dtype=BF)
T, gh, gw = 13, 30, 45
cos, sin = torch.rand(T * gh * gw, D, device="cuda"), torch.rand(T * gh * gw, D, device="cuda")
ksq = ops.qk_layernorm_rope(q, k, gam, bet, gam, bet, cos, sin, 226, 1e-6, q_scale=D ** -0.5 * LOG2E, want_k_sqmax=True)
Sv, Sr, Hc, Dc = 17550, 4050, 16, 128
qc = (torch.randn(B, Sv, Hc * Dc, device="cuda", dtype=BF, generator=g) * (Dc ** -0.25 * LOG2E)).contiguous()
kv = torch.randn(B, Sr, 2 * Hc * Dc, device="cuda", dtype=BF, generator=g)
kc, vc = kv.chunk(2, -1)
kc, ksqc = ops.scale_sqmax(kc, Dc ** -0.25, Hc, Dc)
torch.cuda.synchronize()
for _ in range(iters):
    ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq, bound_proven=True)
    ops.attn_fwd(qc.view(B, Sv, Hc, Dc), kc.view(B, Sr, Hc, Dc), vc.reshape(B, Sr, Hc, Dc), 1.0, log2_scores=True, k_sqmax=ksqc)
torch.cuda.synchronize()
print("done", iters)
