"""A/B of the two MFMA bodies of the bound-centred self-attention at the product shape [2, 17776, 48, 64]: interleaved rounds in ONE
process (cdna_hip_programming.md rule 24), random data, the product's flags (bound proven, tail split).
usage: python tools/attn_body_bench.py [iters] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import inspect as _inspect
from trajectorycrafter_amd import ops as _ops
if "body16" not in _inspect.signature(_ops.attn_fwd).parameters:
    raise SystemExit("this tool compares attention bodies that live in tools/exp/attn_gemm_experiments.patch: run it in the patched copy: "
                     "bash -c '. tools/exp/with_experiments.sh && python3 tools/attn_body_bench.py'")
import torch
from trajectorycrafter_amd import ops
_fused = "--fused" in sys.argv
_args = [a for a in sys.argv[1:] if a != "--fused"]
iters = int(_args[0]) if len(_args) > 0 else 20
rounds = int(_args[1]) if len(_args) > 1 else 6
B, S, H, D = 2, 17776, 48, 64
g = torch.Generator(device="cuda").manual_seed(0)
if _fused:                           # the model's layout: q, k, v are column slices of ONE [B, S, 3 H D] projection output (row stride 18 KB)
    qkv = torch.randn(B, S, 3 * H * D, device="cuda", dtype=torch.bfloat16, generator=g)
    q, k, v = (t.view(B, S, H, D) for t in qkv.chunk(3, -1))
    q.mul_(D ** -0.5 * 1.4426950408889634)
    print("layout: fused QKV views (row stride", q.stride(1) * 2, "bytes)")
else:
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, generator=g) * (D ** -0.5 * 1.4426950408889634)
    k = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, generator=g)
    v = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, generator=g)
ksq = (k.float() ** 2).sum(-1).amax(1).contiguous()
flop = 4.0 * S * S * D * H * B
def t(body16):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq, bound_proven=True, body16=body16)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for _ in range(2):
    t(True); t(False); t(4)
res = {True: [], False: [], 4: []}
for rep in range(rounds):
    a, b, c = t(False), t(True), t(4)
    res[False].append(a); res[True].append(b); res[4].append(c)
    print(f"32x32x16 body {a:.3f} ms ({flop / a / 1e9:.0f} TF)   16x16x32 body {b:.3f} ms ({flop / b / 1e9:.0f} TF)   4-wave body {c:.3f} ms ({flop / c / 1e9:.0f} TF)   "
          f"4-wave vs 32: {100 * (a - c) / a:+.2f} %", flush=True)
med = lambda x: sorted(x)[len(x) // 2]
print(f"median: 32x32x16 {med(res[False]):.3f} ms = {flop / med(res[False]) / 1e9:.0f} TF | 16x16x32 {med(res[True]):.3f} ms = {flop / med(res[True]) / 1e9:.0f} TF"
      f" | 4-wave {med(res[4]):.3f} ms = {flop / med(res[4]) / 1e9:.0f} TF")
