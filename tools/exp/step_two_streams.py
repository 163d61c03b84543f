"""Experiment: the two CFG halves of a denoising step as two B = 1 forwards on two HIP streams (the tail round of every kernel of
one half is filled by the other half's kernels) against the shipped single B = 2 forward.  Alternating blocks in one process.
usage: python tools/exp/step_two_streams.py [steps_per_block] [blocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 3
args = bench.parse([])
dev = torch.device("cuda:0")
pipe = bench.build_models(args, dev)
inp = bench.make_inputs(args, dev, seed=43)
st = pipe.prepare_denoise(prompt=None, height=args.height, width=args.width, num_frames=args.frames, num_inference_steps=50, guidance_scale=6.0, **inp)
side = [torch.cuda.Stream(), torch.cuda.Stream()]
tr = pipe.transformer

def forward_two(t):
    main = torch.cuda.current_stream()
    outs = [None, None]
    for h in (0, 1):
        side[h].wait_stream(main)
        with torch.cuda.stream(side[h]):
            ts = torch.full((1,), t, device=dev, dtype=torch.int64)
            outs[h] = tr(hidden_states=st.latents, encoder_hidden_states=st.prompt_embeds[h:h + 1], timestep=ts,
                         image_rotary_emb=st.image_rotary_emb, return_dict=False, inpaint_latents=st.inpaint_latents[h:h + 1],
                         cross_latents=st.ref_input[h:h + 1])[0]
    for h in (0, 1):
        main.wait_stream(side[h])
    return outs

def step_two(t):
    u, c = forward_two(t)
    st.latents = pipe.scheduler.fused_cfg_step(u, c, st.latents, pipe.guidance_scale, t)

it = [0]
def run(k, two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        t = st.timesteps[it[0] % 50]; it[0] += 1
        step_two(t) if two else pipe.denoise_step(st, t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3

# same result?
lat0 = st.latents.clone()
pipe.denoise_step(st, st.timesteps[0]); a = st.latents.clone()
st.latents = lat0.clone(); step_two(st.timesteps[0]); torch.cuda.synchronize()
print("two-stream step bit-identical to the batched step:", bool(torch.equal(a, st.latents)), flush=True)
run(1, False); run(1, True)
res = {False: [], True: []}
for b in range(blocks):
    for two in (False, True):
        ms = run(n, two); res[two].append(ms)
        print(f"block {b} two_streams={two}: {ms:.1f} ms per step", flush=True)
for two in (False, True):
    print(f"two_streams={two}: median {sorted(res[two])[len(res[two]) // 2]:.1f} ms")
