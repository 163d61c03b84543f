"""Runnable entry points of the denoising hot path (SURVEY §8 rows f2 / e; BASELINE configs[3], configs[4]).

    python -m trajectorycrafter_amd.run generate --model-dir CKPT --transformer-dir CKPT_T --conditioning clip.safetensors \
                                                 --out frames.safetensors [--sampler DDIM_Origin] [--steps 50] [--seed 43]
    python -m torch.distributed.run --nproc-per-node 8 -m trajectorycrafter_amd.run orbits --model-dir ... --clip clip.safetensors \
                                                 --out orbits.safetensors [--radius 1.0] [--variants left_-30,right_30]

`generate` is the part of the reference's `inference.py` / `TrajCrafter.infer_*` that runs after the conditioning stage
(demo.py:121-148): it loads the checkpoint directories exactly as `setup_diffusion` does (demo.py:634-672: the transformer from
`--transformer-dir`, `vae/`, `text_encoder/`, `tokenizer/`, `scheduler/` under `--model-dir`, the sampler from the reference's
table), reads the hand-off file the conditioning stage wrote (`conditioning.save_conditioning`: rendered video, masks, reference
frames, prompt or prompt embeddings), runs `TrajCrafter_Pipeline.__call__` and writes the frames.  `orbits` is
`inference_orbits.py:248-300`: the trajectory variants of one clip (`driver.run_orbits`), one rank per GPU under torchrun.

Outputs are `.safetensors` (`frames` fp32 [B,3,F,H,W] in [0,1], plus timings in the header): video containers (mp4 via
imageio / ffmpeg in the reference's `save_video`) are presentation I/O and stay with the caller.  No CPU fallback: without an
MI355X or without libtcx_hip.so this exits with the `TcxError`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from typing import Optional

import torch

SAMPLERS = ("DDIM_Origin", "DDIM_Cog", "Euler", "Euler A", "DPM++", "PNDM")      # the reference's whole table (demo.py:647-654)
_NOT_BUILT = ()


def make_scheduler(sampler_name: str, model_dir: Optional[str]):
    """demo.py:647-657: the sampler table + `from_pretrained(model_name, subfolder="scheduler")`; the class defaults (the
    CogVideoX-Fun-V1.1-5b-InP values as recalled, scheduler.py) when the directory holds no scheduler_config.json."""
    from . import scheduler as S
    if sampler_name in _NOT_BUILT:
        raise NotImplementedError(f"sampler {sampler_name!r} is in the reference's table but not built on this path; built: {SAMPLERS}")
    if sampler_name not in SAMPLERS:
        raise ValueError(f"unknown sampler {sampler_name!r}; the reference's choices: {SAMPLERS + _NOT_BUILT}")
    cls = {"DDIM_Origin": S.DDIMScheduler, "DDIM_Cog": S.CogVideoXDDIMScheduler, "Euler": S.EulerDiscreteScheduler,
           "Euler A": S.EulerAncestralDiscreteScheduler, "DPM++": S.DPMSolverMultistepScheduler, "PNDM": S.PNDMScheduler}[sampler_name]
    if model_dir and os.path.exists(os.path.join(model_dir, "scheduler", "scheduler_config.json")):
        return cls.from_pretrained(model_dir, subfolder="scheduler")
    return cls()


def load_pipeline(model_dir: str, transformer_dir: Optional[str], sampler_name: str, device: torch.device, need_text_encoder: bool):
    """demo.py:634-672 (`setup_diffusion`) on one MI355X: bf16 weights resident in HBM (the reference's
    `enable_model_cpu_offload` shuttling has nothing to do on 288 GB)."""
    from .models.autoencoder_magvit import AutoencoderKLCogVideoX
    from .models.crosstransformer3d import CrossTransformer3DModel
    from .models.pipeline_trajectorycrafter import TrajCrafter_Pipeline
    bf16 = torch.bfloat16
    tdir = transformer_dir or os.path.join(model_dir, "transformer")
    transformer = CrossTransformer3DModel.from_pretrained(tdir).to(device, bf16).eval()
    vae = AutoencoderKLCogVideoX.from_pretrained(model_dir, subfolder="vae").to(device, bf16).eval()
    tokenizer = text_encoder = None
    if need_text_encoder:
        try:
            from transformers import T5EncoderModel, T5Tokenizer
        except ImportError as e:                             # pragma: no cover
            raise RuntimeError("the hand-off file carries prompt text: `transformers` is needed for the T5 encoder") from e
        tokenizer = T5Tokenizer.from_pretrained(model_dir, subfolder="tokenizer")
        text_encoder = T5EncoderModel.from_pretrained(model_dir, subfolder="text_encoder", torch_dtype=bf16).to(device).eval()
    return TrajCrafter_Pipeline(tokenizer, text_encoder, vae, transformer, make_scheduler(sampler_name, model_dir))


def _device() -> torch.device:
    if not torch.cuda.is_available():
        from ._lib import TcxError
        raise TcxError("trajectorycrafter_amd.run needs an MI355X: the HIP path has no CPU fallback")
    from . import dp
    _, world, local = dp.env_rank()
    if world > 1 and os.environ.get("TCX_BENCH_SINGLE_DEVICE") != "1":
        torch.cuda.set_device(local)
    return torch.device("cuda", torch.cuda.current_device())


def cmd_generate(a) -> int:
    from safetensors.torch import save_file
    from .conditioning import load_conditioning
    dev = _device()
    kw = load_conditioning(a.conditioning, device=dev)
    for name, val in (("num_inference_steps", a.steps), ("guidance_scale", a.guidance_scale), ("height", a.height), ("width", a.width)):
        if val is not None:
            kw[name] = val
    if a.seed is not None and "latents" not in kw:
        kw["generator"] = torch.Generator(device=dev).manual_seed(a.seed)      # demo.py:121
    t0 = time.perf_counter()
    pipe = load_pipeline(a.model_dir, a.transformer_dir, a.sampler, dev, need_text_encoder=kw.get("prompt") is not None)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    if a.global_seed is not None:
        torch.manual_seed(a.global_seed)                    # the posterior sample of the reference latents draws from the global RNG
    t0 = time.perf_counter()
    frames = pipe(output_type="pt", **kw).videos
    torch.cuda.synchronize()
    t_run = time.perf_counter() - t0
    tm = pipe.timings()
    meta = {"format": "trajectorycrafter-frames/1", "sampler": a.sampler, "load_seconds": f"{t_load:.3f}", "call_seconds": f"{t_run:.3f}",
            "denoise_seconds": f"{tm['denoise_s']:.3f}", "decode_seconds": f"{tm['decode_s']:.3f}", "shape": json.dumps(list(frames.shape))}
    save_file({"frames": frames.float().cpu().contiguous()}, a.out, metadata=meta)
    print(json.dumps(dict(meta, out=a.out)), flush=True)
    return 0


def cmd_orbits(a) -> int:
    from safetensors import safe_open
    from safetensors.torch import save_file
    from . import dp
    from .driver import ORBIT_VARIANTS, run_orbits
    from .models.utils import Warper
    rank, world, _ = dp.env_rank()
    dev = _device()
    if world > 1:
        dp.init_distributed(os.environ.get("TCX_DIST_BACKEND", "nccl"))
    clip, extra = {}, {}
    with safe_open(a.clip, framework="pt") as f:
        header = f.metadata() or {}
        for k in f.keys():
            clip[k] = f.get_tensor(k).to(dev)
    for need in ("frames", "depths"):
        if need not in clip:
            raise ValueError(f"{a.clip}: needs tensors `frames` [F,3,H,W] in [-1,1] and `depths` [F,1,H,W] (+ prompt_embeds / "
                             f"negative_prompt_embeds or `prompt` in the header); missing {need!r}")
    variants = ORBIT_VARIANTS
    if a.variants:
        table = dict(ORBIT_VARIANTS)
        unknown = [n for n in a.variants.split(",") if n not in table]
        if unknown:
            raise ValueError(f"unknown variants {unknown}; known: {sorted(table)}")
        variants = tuple((n, table[n]) for n in a.variants.split(","))
    if a.traj_txt:                                        # reference --camera traj --traj_txt FILE (demo.py:566-573), one variant per file
        from .driver import read_traj_txt
        files = a.traj_txt.split(",")
        trajs = tuple((os.path.splitext(os.path.basename(f))[0], read_traj_txt(f)) for f in files)
        variants = trajs if not a.variants else variants + trajs
    prompt = header.get("prompt")
    if "prompt_embeds" in clip:
        extra = dict(prompt_embeds=clip["prompt_embeds"].to(torch.bfloat16),
                     negative_prompt_embeds=clip["negative_prompt_embeds"].to(torch.bfloat16) if "negative_prompt_embeds" in clip else None)
        prompt = None
    pipe = load_pipeline(a.model_dir, a.transformer_dir, a.sampler, dev, need_text_encoder=prompt is not None)
    t0 = time.perf_counter()
    frames = run_orbits(pipe, Warper(device=str(dev)), clip["frames"], clip["depths"], variants=variants, radius=a.radius,
                        radius_scale=a.radius_scale, K=clip.get("K"), sample_size=(a.height or 384, a.width or 672), prompt=prompt,
                        negative_prompt=header.get("negative_prompt"), guidance_scale=a.guidance_scale if a.guidance_scale is not None else 6.0,
                        num_inference_steps=a.steps or 50, seed=43 if a.seed is None else a.seed, mask=not a.no_mask, **extra)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        meta = {"format": "trajectorycrafter-frames/1", "variants": json.dumps([n for n, _ in variants]), "world_size": str(world),
                "seconds": f"{dt:.3f}", "shape": json.dumps(list(frames.shape))}
        save_file({"frames": frames.float().cpu().contiguous()}, a.out, metadata=meta)
        print(json.dumps(dict(meta, out=a.out)), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0


def parse(argv=None):
    ap = argparse.ArgumentParser(prog="python -m trajectorycrafter_amd.run", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("generate", "orbits"):
        p = sub.add_parser(name)
        p.add_argument("--model-dir", required=True, help="CogVideoX-Fun checkpoint dir: vae/ scheduler/ [text_encoder/ tokenizer/]")
        p.add_argument("--transformer-dir", default=None, help="CrossTransformer3DModel dir (default: MODEL_DIR/transformer)")
        p.add_argument("--out", required=True, help="output .safetensors (tensor `frames`)")
        p.add_argument("--sampler", default="DDIM_Origin", help=f"one of {SAMPLERS} (reference --sampler_name)")
        p.add_argument("--steps", type=int, default=None, help="denoising steps (reference --diffusion_inference_steps, 50)")
        p.add_argument("--guidance-scale", type=float, default=None, help="reference --diffusion_guidance_scale, 6.0")
        p.add_argument("--seed", type=int, default=None, help="reference --seed, 43")
        p.add_argument("--height", type=int, default=None)
        p.add_argument("--width", type=int, default=None)
    g = sub.choices["generate"]
    g.add_argument("--conditioning", required=True, help="hand-off file written by conditioning.save_conditioning")
    g.add_argument("--global-seed", type=int, default=None, help="torch.manual_seed before the call (pins the posterior sample)")
    o = sub.choices["orbits"]
    o.add_argument("--clip", required=True, help=".safetensors with `frames` [F,3,H,W] in [-1,1], `depths` [F,1,H,W], prompt embeddings")
    o.add_argument("--radius", type=float, default=1.0, help="reference --radius")
    o.add_argument("--radius-scale", type=float, default=1.0, help="reference --radius_scale")
    o.add_argument("--variants", default=None, help="comma-separated subset of the reference's variant names (default: all eight)")
    o.add_argument("--traj-txt", default=None, help="comma-separated trajectory files (reference --camera traj --traj_txt: theta / phi / r "
                                                     "key lines, test/trajs/loop1.txt); alone: only these; with --variants: in addition")
    o.add_argument("--no-mask", action="store_true", help="reference opts.mask = False (inference_orbits.py sets True)")
    return ap.parse_args(argv)


def main(argv=None) -> int:
    a = parse(argv)
    return cmd_generate(a) if a.cmd == "generate" else cmd_orbits(a)


if __name__ == "__main__":
    sys.exit(main())
