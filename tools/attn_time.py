"""Wall time of the product's self-attention launch (fused-QKV layout [2,17776,48,64], bound proven, tail split) and cross-attention
launch, HIP events over back-to-back launches after a 2 s warm-up.  usage: python tools/attn_time.py [iters]   (TCX_LIB picks the library)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectorycrafter_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
BF, LOG2E = torch.bfloat16, 1.4426950408889634
g = torch.Generator(device="cuda").manual_seed(0)
B, S, H, D = 2, 17776, 48, 64
qkv = torch.randn(B, S, 3 * H * D, device="cuda", dtype=BF, generator=g)
q, k, v = (t.view(B, S, H, D) for t in qkv.chunk(3, -1))
gam, bet = torch.ones(D, device="cuda", dtype=BF), torch.zeros(D, device="cuda", dtype=BF)
cos, sin = torch.rand(17550, D, device="cuda"), torch.rand(17550, D, device="cuda")
ksq = ops.qk_layernorm_rope(q, k, gam, bet, gam, bet, cos, sin, 226, 1e-6, q_scale=D ** -0.5 * LOG2E, want_k_sqmax=True)
Sv, Sr, Hc, Dc = 17550, 4050, 16, 128
qc = (torch.randn(B, Sv, Hc * Dc, device="cuda", dtype=BF, generator=g) * (Dc ** -0.25 * LOG2E)).contiguous()
kv = torch.randn(B, Sr, 2 * Hc * Dc, device="cuda", dtype=BF, generator=g)
kc, vc = kv.chunk(2, -1)
kc, ksqc = ops.scale_sqmax(kc, Dc ** -0.25, Hc, Dc)
self_ = lambda **kw: ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq, **kw)
cross = lambda: ops.attn_fwd(qc.view(B, Sv, Hc, Dc), kc.view(B, Sr, Hc, Dc), vc.reshape(B, Sr, Hc, Dc), 1.0, log2_scores=True, k_sqmax=ksqc)
def t(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t0 = time.time()
while time.time() - t0 < 2.0: self_(bound_proven=True)
torch.cuda.synchronize()
a = [t(lambda: self_(bound_proven=True), iters) for _ in range(3)]
b = [t(lambda: self_(), iters) for _ in range(2)]
e = [t(lambda: ops.attn_fwd(q, k, v, 1.0, log2_scores=True), iters) for _ in range(2)]
c = [t(cross, iters) for _ in range(2)]
fl, flc = 4.0 * S * S * D * H * B, 4.0 * Sv * Sr * Dc * Hc * B
print(f"lib {os.environ.get('TCX_LIB', 'in-tree')}: self proven {min(a):.3f} ms ({fl / min(a) / 1e9:.0f} TF) | tested {min(b):.3f} | exact {min(e):.3f} ({fl / min(e) / 1e9:.0f} TF) | cross {min(c):.3f} ms ({flc / min(c) / 1e9:.0f} TF)")
