"""In-process A/B of a module-level switch on the FULL denoising step (the bench's step: 42-layer model at 49f 480x720, CFG batch 2):
alternating blocks of steps with the switch off / on, so that both arms see the same box, clock history and data.
usage: python tools/step_ab.py [steps_per_block] [blocks] [switch]      switch: cross_kv (default; model.cache_cross_kv: reuse of the
cross-attention K / V across steps) | unproven | exact (model.set_softmax_path: the attention's other softmax loops, as in bench.py's
bracket) | body16 (ops.ATTN_BODY16_DEFAULT — only in the patched copy: bash -c '. tools/exp/with_experiments.sh && python3 tools/step_ab.py 5 6 body16')"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from trajectorycrafter_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
switch = sys.argv[3] if len(sys.argv) > 3 else "cross_kv"
if switch == "body16" and not hasattr(ops, "ATTN_BODY16_DEFAULT"):
    raise SystemExit("the 16x16x32 body lives in tools/exp/attn_gemm_experiments.patch: bash -c '. tools/exp/with_experiments.sh && python3 tools/step_ab.py 5 6 body16'")
args = bench.parse([])
dev = torch.device("cuda:0")
pipe = bench.build_models(args, dev)
inp = bench.make_inputs(args, dev, seed=43)
st = pipe.prepare_denoise(prompt=None, height=args.height, width=args.width, num_frames=args.frames, num_inference_steps=50, guidance_scale=6.0, **inp)
it = [0]
def steps(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.attn_timing_start()
    for _ in range(k):
        pipe.denoise_step(st, st.timesteps[it[0] % 50]); it[0] += 1
    a = ops.attn_timing_stop()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, a[64]["ms"] / a[64]["n"]
steps(2)
res = {False: [], True: []}
for b in range(blocks):
    for flag in (False, True):
        if switch == "body16":
            ops.ATTN_BODY16_DEFAULT = flag
        elif switch in ("unproven", "exact"):
            pipe.transformer.set_softmax_path(switch if flag else "auto")
        else:
            pipe.transformer.cache_cross_kv = flag
            if flag:
                steps(1)                                   # fill the cache outside the timed block
        ms, att = steps(n)
        res[flag].append((ms, att))
        print(f"block {b} {switch}={flag}: {ms:.1f} ms per step, self-attention {att:.3f} ms per launch", flush=True)
for flag in (False, True):
    print(f"{switch}={flag}: median step {sorted(x[0] for x in res[flag])[len(res[flag]) // 2]:.1f} ms, attention {sorted(x[1] for x in res[flag])[len(res[flag]) // 2]:.3f} ms")
