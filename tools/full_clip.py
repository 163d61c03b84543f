"""ONE real 50-step 480x720 clip through `TrajCrafter_Pipeline.__call__` (denoise loop + VAE decode, the product entry point) timed
wall-clock, next to bench.py's composed figure `50 * step + decode` measured in the same process on the same box — the check that
the bench's `value` formula describes a clip that actually runs.  Writes gpurun_out/<tag>_full_clip.json.
usage: python tools/full_clip.py [tag]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
args = bench.parse(["--steps", "20", "--warmup", "5"])
dev = torch.device("cuda:0")
pipe = bench.build_models(args, dev)
inp = bench.make_inputs(args, dev, seed=43)
kw = dict(prompt=None, height=args.height, width=args.width, num_frames=args.frames, guidance_scale=6.0, **inp)


def sync_time(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


# (1) the bench's way: warm-up, K consecutive steps of one trajectory, one decode
st = pipe.prepare_denoise(num_inference_steps=50, **kw)
for i in range(args.warmup):
    pipe.denoise_step(st, st.timesteps[i])
t_steps, _ = sync_time(lambda: [pipe.denoise_step(st, st.timesteps[args.warmup + i]) for i in range(args.steps)])
step_s = t_steps / args.steps
pipe.decode_latents(st.latents)
decode_s, _ = sync_time(lambda: pipe.decode_latents(st.latents))
composed = 50 * step_s + decode_s
print(f"composed: step {1e3 * step_s:.1f} ms, decode {1e3 * decode_s:.1f} ms -> {composed:.3f} s per clip", flush=True)

# (2) one real clip: prepare + 50 steps + decode, fp32 frames on the device (output_type='pt'; the reference's .cpu() of the
#     203 MB frame tensor is PCIe time, reported separately)
wall, out = sync_time(lambda: pipe(num_inference_steps=50, output_type="pt", **kw).videos)
tm = pipe.timings()
t_cpu, host = sync_time(lambda: out.cpu())
assert out.shape == (1, 3, 49, 480, 720) and torch.isfinite(out).all() and 0 <= float(out.min()) and float(out.max()) <= 1
rec = {"workload": "configs[2]: 50-step CFG DDIM + VAE decode, 49f 480x720, 42-layer random-init model, 1 MI355X",
       "bench_formula": {"ms_per_step": 1e3 * step_s, "decode_ms": 1e3 * decode_s, "clip_seconds": composed, "timed_steps": args.steps},
       "real_clip": {"wall_seconds_call": wall, "denoise_seconds_events": tm["denoise_s"], "decode_seconds_events": tm["decode_s"],
                     "prepare_and_host_seconds": wall - tm["denoise_s"] - tm["decode_s"], "frames_to_host_seconds_pcie": t_cpu},
       "real_over_formula": wall / composed, "loop_plus_decode_over_formula": (tm["denoise_s"] + tm["decode_s"]) / composed}
print(json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"{tag}_full_clip.json"), "w") as f:
    json.dump(rec, f, indent=1)
