"""A/B of the attention tail split at the product shape [2, 17776, 48, 64]: interleaved timing of split / single-pass launches.
usage: python tools/attn_split_bench.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectorycrafter_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, S, H, D = 2, 17776, 48, 64
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, generator=g) * (D ** -0.5 * 1.4426950408889634)
k = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, generator=g)
v = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, generator=g)
ksq = (k.float() ** 2).sum(-1).amax(1).contiguous()
flop = 4.0 * S * S * D * H * B
def t(split):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq, split_tail=split)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for _ in range(2):
    t(True); t(False)
for rep in range(4):
    a, b = t(True), t(False)
    print(f"split {a:.3f} ms ({flop / a / 1e9:.0f} TF)   single-pass {b:.3f} ms ({flop / b / 1e9:.0f} TF)   gain {100 * (b - a) / b:.2f} %", flush=True)
