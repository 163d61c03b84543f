"""`python -m trajectorycrafter_amd.run generate` as a PROGRAM (BASELINE configs[4] minus the conditioning stage and
real weights): a checkpoint directory in the reference's layout (transformer/ + vae/ + scheduler/scheduler_config.json, sharded
safetensors) and a conditioning hand-off file go in, a frames file comes out; equal, bit for bit, to calling the pipeline in
process.  Also the `DDIM_Cog` sampler through the same entry point and the refusal of samplers that are not built."""
import ast
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16


def _run(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return subprocess.run([sys.executable, "-m", "trajectorycrafter_amd.run", *args], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_generate_entry_point_on_a_checkpoint_dir(golden, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from safetensors.torch import load_file
    from tests.test_models_gpu import _weights
    from trajectorycrafter_amd.conditioning import load_conditioning, save_conditioning
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    from trajectorycrafter_amd.models.pipeline_trajectorycrafter import TrajCrafter_Pipeline
    from trajectorycrafter_amd.scheduler import CogVideoXDDIMScheduler, DDIMScheduler
    tp, _ = golden("pipeline_tiny.safetensors")
    tt, mt = golden("transformer_tiny.safetensors")
    tv, mv = golden("vae_tiny.safetensors")
    tr = CrossTransformer3DModel(**ast.literal_eval(mt["config"]))
    tr.load_state_dict(_weights(tt), strict=True)
    vae = AutoencoderKLCogVideoX(**ast.literal_eval(mv["config"]))
    vae.load_state_dict(_weights(tv), strict=True)
    ckpt = tmp_path / "ckpt"
    tr.to(BF).save_pretrained(str(ckpt / "transformer"), max_shard_size=200_000)          # sharded + index, like the 5B checkpoint
    vae.to(BF).save_pretrained(str(ckpt / "vae"))
    os.makedirs(ckpt / "scheduler")
    (ckpt / "scheduler" / "scheduler_config.json").write_text(json.dumps(
        {"_class_name": "DDIMScheduler", "num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
         "prediction_type": "v_prediction", "timestep_spacing": "trailing", "rescale_betas_zero_snr": True, "snr_shift_scale": 1.0}))
    cond = str(tmp_path / "clip0.safetensors")
    save_conditioning(cond, cond_video=tp["video"], cond_masks=tp["mask_video"], frames_ref=tp["reference"], prompt_embeds=tp["prompt_embeds"],
                      negative_prompt_embeds=tp["negative_prompt_embeds"], latents=tp["latents0"], height=32, width=48, num_frames=9,
                      num_inference_steps=2, guidance_scale=6.0)
    out = str(tmp_path / "frames.safetensors")
    r = _run(["generate", "--model-dir", str(ckpt), "--conditioning", cond, "--out", out, "--global-seed", "5"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["out"] == out and json.loads(line["shape"]) == [1, 3, 9, 32, 48] and line["sampler"] == "DDIM_Origin"
    got = load_file(out)["frames"]
    assert got.dtype == torch.float32 and got.shape == (1, 3, 9, 32, 48) and float(got.min()) >= 0 and float(got.max()) <= 1

    # the same call in process: bit-identical
    dev = torch.device("cuda:0")
    pipe = TrajCrafter_Pipeline(None, None, vae.to(dev, BF).eval(), tr.to(dev, BF).eval(), DDIMScheduler())
    torch.manual_seed(5)
    want = pipe(output_type="pt", **load_conditioning(cond, device=dev)).videos
    assert torch.equal(got, want.float().cpu())

    # DDIM_Cog through the entry point == the pipeline with CogVideoXDDIMScheduler; close to DDIM_Origin (same update, other
    # rounding points) but not the same bits
    out2 = str(tmp_path / "frames_cog.safetensors")
    r2 = _run(["generate", "--model-dir", str(ckpt), "--conditioning", cond, "--out", out2, "--sampler", "DDIM_Cog", "--global-seed", "5"])
    assert r2.returncode == 0, r2.stdout[-1500:] + r2.stderr[-3000:]
    got2 = load_file(out2)["frames"]
    pipe2 = TrajCrafter_Pipeline(None, None, pipe.vae, pipe.transformer, CogVideoXDDIMScheduler())
    torch.manual_seed(5)
    want2 = pipe2(output_type="pt", **load_conditioning(cond, device=dev)).videos
    assert torch.equal(got2, want2.float().cpu())
    assert not torch.equal(got2, got) and float((got2 - got).abs().mean()) < 0.02

    # the other built samplers of the reference's table run through the entry point too (a different clip each)
    out3 = str(tmp_path / "frames_dpm.safetensors")
    r3 = _run(["generate", "--model-dir", str(ckpt), "--conditioning", cond, "--out", out3, "--sampler", "DPM++", "--global-seed", "5"])
    assert r3.returncode == 0, r3.stdout[-1500:] + r3.stderr[-3000:]
    got3 = load_file(out3)["frames"]
    assert got3.shape == got.shape and torch.isfinite(got3).all() and not torch.equal(got3, got)
    r4 = _run(["generate", "--model-dir", str(ckpt), "--conditioning", cond, "--out", out, "--sampler", "LCM"])
    assert r4.returncode != 0 and "unknown sampler" in r4.stderr


def test_orbits_entry_point_single_process(golden, tmp_path):
    """`python -m trajectorycrafter_amd.run orbits` (world size 1 = the reference's sequential loop over the variants,
    inference_orbits.py:285-300) on a checkpoint directory + a clip file (frames, depths, K, prompt embeddings): equal to
    `driver.run_orbits` called in process on the same inputs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from safetensors.torch import load_file, save_file
    from tests.orbit_rank_worker import build, clip
    from trajectorycrafter_amd.driver import ORBIT_VARIANTS, run_orbits
    dev = torch.device("cuda:0")
    pipe, warper = build(dev)
    ckpt = tmp_path / "ckpt"
    pipe.transformer.save_pretrained(str(ckpt / "transformer"))
    pipe.vae.save_pretrained(str(ckpt / "vae"))
    frames, depths, K, pe, ne = clip(dev)
    clip_file = str(tmp_path / "clip.safetensors")
    save_file({"frames": frames.cpu(), "depths": depths.cpu(), "K": K.cpu(), "prompt_embeds": pe.cpu(), "negative_prompt_embeds": ne.cpu()}, clip_file)
    out = str(tmp_path / "orbits.safetensors")
    r = _run(["orbits", "--model-dir", str(ckpt), "--clip", clip_file, "--out", out, "--variants", "right_30,left_-45", "--radius", "0.6",
              "--steps", "2", "--height", "32", "--width", "48"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert json.loads(line["variants"]) == ["right_30", "left_-45"] and line["world_size"] == "1"
    got = load_file(out)["frames"]
    table = dict(ORBIT_VARIANTS)
    want = run_orbits(pipe, warper, frames, depths, variants=(("right_30", table["right_30"]), ("left_-45", table["left_-45"])), radius=0.6, K=K,
                      sample_size=(32, 48), prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=2, seed=43, mask=True)
    assert got.shape == (2, 3, 9, 32, 48) and torch.equal(got, want.float().cpu())
    bad = _run(["orbits", "--model-dir", str(ckpt), "--clip", clip_file, "--out", out, "--variants", "sideways"])
    assert bad.returncode != 0 and "unknown variants" in bad.stderr
    # a trajectory file (reference --camera traj --traj_txt, demo.py:566-573): one more variant, named after the file
    traj = tmp_path / "sweep.txt"
    traj.write_text("0 4 8 4 0\n0 -10 -25\n0 0.1 0.2 0.1 0\n")
    out2 = str(tmp_path / "traj.safetensors")
    r2 = _run(["orbits", "--model-dir", str(ckpt), "--clip", clip_file, "--out", out2, "--traj-txt", str(traj), "--radius", "0.6",
               "--steps", "2", "--height", "32", "--width", "48"])
    assert r2.returncode == 0, r2.stdout[-1500:] + r2.stderr[-3000:]
    assert json.loads(json.loads(r2.stdout.strip().splitlines()[-1])["variants"]) == ["sweep"]
    from trajectorycrafter_amd.driver import read_traj_txt
    want2 = run_orbits(pipe, warper, frames, depths, variants=(("sweep", read_traj_txt(str(traj))),), radius=0.6, K=K, sample_size=(32, 48),
                       prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=2, seed=43, mask=True)
    got2 = load_file(out2)["frames"]
    assert got2.shape == (1, 3, 9, 32, 48) and torch.equal(got2, want2.float().cpu()) and not torch.equal(got2[0], got[0])
