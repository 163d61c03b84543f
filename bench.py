#!/usr/bin/env python
"""Headline benchmark: denoised video-latents/sec at 49 frames, 480x720 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one whole clip of BASELINE configs[2] ("Full 50-step DDIM, 49-frame 480x720, random-init
weights, 1 MI355X + MAGViT VAE decode"): 50 denoising steps with classifier-free guidance (each a B=2
forward of the 42-block / 6.1 B-parameter CrossTransformer3DModel + the fused CFG/DDIM update) followed
by the VAE decode to 49 frames.  Nothing is skipped or cached inside the timed region.  Inputs
(latents, prompt embeddings, inpaint / reference latents) are synthetic and already resident in HBM.
With N > 1 every rank denoises its own independent trajectory (weak scaling, no data-path collective)
and one RCCL all-gather reassembles the decoded frames at the end of every step.

Rank 0 prints ONE JSON line (see the driver contract) including
  "roofline"     — the dominant kernel (self-attention, tcx_attn_fwd D=64): algorithmic FLOP per launch
                   / average launch duration measured live with HIP events on the launch stream;
  "cpu_baseline" — the oracle (CPU port of the reference) timed on the host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16 = torch.bfloat16
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md; 2:1-sparsity figures never used)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1, help="timed clips per rank")
    ap.add_argument("--warmup", type=int, default=1, help="untimed warm-up clips per rank")
    ap.add_argument("--denoise-steps", type=int, default=50)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=720)
    ap.add_argument("--frames", type=int, default=49)
    ap.add_argument("--layers", type=int, default=42, help="(debug) fewer layers => NOT the benchmark config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="(debug) skip the VAE decode => NOT the benchmark config")
    return ap.parse_args()


def build_models(args, device):
    from trajectorycrafter_amd import init_weights as iw
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    from trajectorycrafter_amd.models.pipeline_trajectorycrafter import TrajCrafter_Pipeline
    from trajectorycrafter_amd.scheduler import DDIMScheduler

    tcfg = dict(iw.TRANSFORMER_5B, num_layers=args.layers)
    with torch.device("meta"):
        tr = CrossTransformer3DModel(**tcfg)
        vae = AutoencoderKLCogVideoX()
    tr.load_state_dict(iw.random_state_dict(iw.transformer_param_shapes(dict(tr.config)), seed=0, dtype=BF16, device=device),
                       strict=True, assign=True)
    vsd = iw.random_state_dict(iw.vae_param_shapes(dict(vae.config), decoder=True, encoder=True), seed=1, dtype=BF16,
                               device=device)
    vae.load_state_dict(vsd, strict=True, assign=True)
    tr.eval(), vae.eval()
    return TrajCrafter_Pipeline(None, None, vae, tr, DDIMScheduler()), tcfg


def make_inputs(args, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    T = (args.frames - 1) // 4 + 1
    h, w = args.height // 8, args.width // 8
    rn = lambda *s: torch.randn(*s, device=device, dtype=BF16, generator=g)
    return dict(latents=rn(1, T, 16, h, w), prompt_embeds=rn(1, 226, 4096), negative_prompt_embeds=rn(1, 226, 4096),
                inpaint_latents=rn(2, T, 17, h, w), ref_latents=rn(2, 3, 16, h, w))


def cpu_baseline(args, threads: int = 16, attn_heads: int = 4, cross_heads: int = 2):
    """Oracle (CPU port of the reference, fp32) on the host cores, bounded to ~10-30 s: at the full
    480x720 token count, B=1, time (a) every row-wise / GEMM piece of ONE CogVideoXBlock, (b) its joint
    attention on `attn_heads` of the 48 heads, (c) the GEMMs + LayerNorms of ONE PerceiverCrossAttention and
    (d) its attention on `cross_heads` of the 16 heads; heads are independent, so (b) and (d) scale linearly.
    A clip is extrapolated as 2*steps forwards x (42 blocks + 21 cross layers).  Reported baseline, not the target."""
    import torch.nn.functional as F
    from oracle import diffusers_restated as dr
    from oracle import transformer as otr
    from oracle.pipeline import prepare_rotary
    from oracle.prec import Prec
    from trajectorycrafter_amd import init_weights as iw

    threads = max(1, min(threads, os.cpu_count() or 1))        # the box's CPU share for one GPU is 16 cores
    torch.set_num_threads(threads)
    cfg = dict(otr.DEFAULT_CONFIG, **dict(iw.TRANSFORMER_5B, num_layers=2))
    sd = iw.random_state_dict(iw.transformer_param_shapes(cfg), seed=0)
    T = (args.frames - 1) // 4 + 1
    gh, gw = args.height // 16, args.width // 16
    Sv, S, Sr, D = T * gh * gw, T * gh * gw + 226, 3 * gh * gw, 3072
    g = torch.Generator().manual_seed(0)
    p = Prec("fp32")
    x = torch.randn(1, S, D, generator=g)
    temb, ref = torch.randn(1, 512, generator=g), torch.randn(1, Sr, D, generator=g)
    cos, sin = prepare_rotary(args.height, args.width, T, 2, 64)
    pre = "transformer_blocks.0."
    tick = time.perf_counter
    # (a) block without attention: 2 x LayerNormZero, qkv + out projections, q/k LayerNorm + RoPE, FFN, gated residuals
    t0 = tick()
    n1, e1, gate, egate = dr.layer_norm_zero(p, sd, pre + "norm1.", x[:, 226:], x[:, :226], temb, 1e-5)
    h = torch.cat([e1, n1], 1)
    q, k, v = (p.linear(h, sd[pre + f"attn1.{n}.weight"], sd[pre + f"attn1.{n}.bias"]).view(1, S, 48, 64).transpose(1, 2)
               for n in ("to_q", "to_k", "to_v"))
    q = p.layer_norm(q, sd[pre + "attn1.norm_q.weight"], sd[pre + "attn1.norm_q.bias"], 1e-6)
    k = p.layer_norm(k, sd[pre + "attn1.norm_k.weight"], sd[pre + "attn1.norm_k.bias"], 1e-6)
    q = torch.cat([q[:, :, :226], dr.apply_rotary_emb(q[:, :, 226:], cos, sin)], 2)
    k = torch.cat([k[:, :, :226], dr.apply_rotary_emb(k[:, :, 226:], cos, sin)], 2)
    o = p.linear(v.transpose(1, 2).reshape(1, S, D), sd[pre + "attn1.to_out.0.weight"], sd[pre + "attn1.to_out.0.bias"])
    x2 = x + torch.cat([egate.expand(1, 226, D), gate.expand(1, Sv, D)], 1) * o
    n2, e2, gate, egate = dr.layer_norm_zero(p, sd, pre + "norm2.", x2[:, 226:], x2[:, :226], temb, 1e-5)
    ff = dr.feed_forward(p, sd, pre + "ff.", torch.cat([e2, n2], 1))
    x3 = x2 + torch.cat([egate.expand(1, 226, D), gate.expand(1, Sv, D)], 1) * ff
    t_rows = tick() - t0
    # (b) joint attention, `attn_heads` heads
    t0 = tick()
    dr.sdpa(p, q[:, :attn_heads], k[:, :attn_heads], v[:, :attn_heads], 0.125)
    t_attn = (tick() - t0) * 48 / attn_heads
    # (c) cross layer without attention
    cp = "perceiver_cross_attention.0."
    t0 = tick()
    xn = p.layer_norm(ref, sd[cp + "norm1.weight"], sd[cp + "norm1.bias"], 1e-5)
    ln = p.layer_norm(x3[:, 226:], sd[cp + "norm2.weight"], sd[cp + "norm2.bias"], 1e-5)
    cq = p.linear(ln, sd[cp + "to_q.weight"]).view(1, Sv, 16, 128).transpose(1, 2) * 128 ** -0.25
    ck, cv = (t.view(1, Sr, 16, 128).transpose(1, 2) for t in p.linear(xn, sd[cp + "to_kv.weight"]).chunk(2, -1))
    ck = ck * 128 ** -0.25
    p.linear(cq.transpose(1, 2).reshape(1, Sv, 2048), sd[cp + "to_out.weight"])
    t_crows = tick() - t0
    # (d) cross attention, `cross_heads` heads
    t0 = tick()
    dr.sdpa(p, cq[:, :cross_heads], ck[:, :cross_heads], cv[:, :cross_heads], 1.0)
    t_cattn = (tick() - t0) * 16 / cross_heads
    t_block, t_cross = t_rows + t_attn, t_crows + t_cattn
    clip_s = 2 * args.denoise_steps * (42 * t_block + 21 * t_cross)
    return {"value": 1.0 / clip_s, "unit": "video-latents/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32, {args.frames}f {args.height}x{args.width} (S={S}), B=1: CogVideoXBlock = rows/GEMMs {t_rows:.2f} s + "
                      f"attention {t_attn:.2f} s ({attn_heads}/48 heads timed, x{48 // attn_heads}); PerceiverCrossAttention = rows/GEMMs "
                      f"{t_crows:.2f} s + attention {t_cattn:.2f} s ({cross_heads}/16 heads timed); clip extrapolated as "
                      f"{2 * args.denoise_steps} forwards x (42 blocks + 21 cross layers); embeds and VAE decode excluded",
            "extrapolated_clip_seconds": clip_s}


def main():
    args = parse()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal knobs (never set by the driver): run several ranks on ONE GPU with gloo to exercise the N > 1 control flow
    if os.environ.get("TCX_BENCH_SINGLE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    from trajectorycrafter_amd import dp, ops

    if world > 1:
        dp.init_distributed(os.environ.get("TCX_DIST_BACKEND", "nccl"))
    pipe, tcfg = build_models(args, device)
    inp = make_inputs(args, device, seed=43 + rank)            # one independent trajectory per rank (seeds 43..50)

    def one_clip():
        out = pipe(prompt=None, height=args.height, width=args.width, num_frames=args.frames,
                   num_inference_steps=args.denoise_steps, guidance_scale=6.0,
                   output_type="latent" if args.no_decode else "pt", **inp).videos
        return dp.all_gather_cat(out) if world > 1 else out     # the single RCCL all-gather of the path

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_clip()
    ops.attn_timing_start()                                     # HIP events around every tcx_attn_fwd launch
    barrier()
    t0 = time.perf_counter()
    denoise_s = decode_s = 0.0
    for _ in range(args.steps):
        out = one_clip()
    barrier()
    elapsed = time.perf_counter() - t0
    tm = pipe.timings()
    attn = ops.attn_timing_stop()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape[0] == world and torch.isfinite(out.float()).all()

    if rank == 0:
        S = ((args.frames - 1) // 4 + 1) * (args.height // 16) * (args.width // 16) + 226
        flop_per_launch = 4.0 * S * S * 3072 * 2                # 4*S^2*D per sample x B=2 (CFG)
        sa = attn.get(64, {"n": 0, "ms": 0.0})
        avg_ms = sa["ms"] / max(sa["n"], 1)
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        fwd_flop = {(49, 480, 720): 3.5585e14, (49, 384, 672): 2.3387e14}.get((args.frames, args.height, args.width))
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r1_attn_pmc.json")        # PMC counters cannot be read inside the timed run:
        if os.path.exists(pmc) and (args.frames, args.height, args.width) == (49, 480, 720):   # committed rocprofv3 --pmc result
            with open(pmc) as f:
                traffic = json.load(f).get("traffic_bytes_per_launch")
        benchmark_config = (args.denoise_steps, args.height, args.width, args.frames, args.layers, args.no_decode) == (50, 480, 720, 49, 42, False)
        rec = {
            "metric": "denoised video-latents/sec (49f, 480x720)", "value": world * args.steps / elapsed,
            "unit": "video-latents/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[2]: full 50-step DDIM (CFG 6, B=2 per step) + VAE decode, 49f 480x720, random-init 42-layer "
                                    "CrossTransformer3D (the 6.1 B-param 5B model); one independent trajectory per GPU") if benchmark_config else
                                   (f"DEBUG (not the benchmark config): {args.denoise_steps}-step DDIM, {args.frames}f {args.height}x{args.width}, "
                                    f"{args.layers} layers, decode={not args.no_decode}"),
                       "frames": args.frames, "height": args.height, "width": args.width, "denoise_steps": args.denoise_steps,
                       "layers": args.layers, "vae_decode": not args.no_decode, "global_batch_clips": world,
                       "parallelism": f"dp{world}", "last_clip_denoise_s": tm["denoise_s"], "last_clip_decode_s": tm["decode_s"],
                       "transformer_mfma_frac": (None if fwd_flop is None else
                                                 2 * args.denoise_steps * fwd_flop * (args.layers / 42) / tm["denoise_s"] / 1e12 / PEAK_BF16_TFLOPS)},
            "roofline": {"kernel": "tcx_attn_fwd<64> (joint self-attention, 42 launches per step)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                         "traffic": traffic, "traffic_source": "profiles/r1_attn_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)",
                         "launches": sa["n"], "avg_launch_ms": avg_ms,
                         "algorithmic_flop_per_launch": flop_per_launch},
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
