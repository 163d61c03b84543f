"""Thin driver: point-cloud render -> conditioning -> denoise -> frames, all on one MI355X.

Mirrors the data flow of the reference's `TrajCrafter.infer_*` (demo.py:75-148) from the point where frames, depths,
camera poses and the prompt embedding exist: those come from the reference's depth estimator, pose generator and
captioner, which are control plane / other models and out of scope.  Everything between is on the GPU:
`Warper.forward_warp` (one launch for the clip), the three resizes (torch ops on device memory), the HIP VAE encoder inside
`TrajCrafter_Pipeline.__call__`, the denoise loop and the decode.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import dp
from .models.utils import Warper

# The eight target-pose variants of one clip the reference renders and denoises one after the other
# (inference_orbits.py:258-283): (name, [d_theta, d_phi, d_r / radius, d_x, d_y]); d_r is multiplied by `radius` (:242).
ORBIT_VARIANTS = (
    ("left_-30", (0, -30, 1, 0, 0)), ("right_30", (0, 30, 1, 0, 0)), ("top_30", (30, 0, 1, 0, 0)),
    ("left_-45", (0, -45, 1, 0, 0)), ("right_45", (0, 45, 1, 0, 0)), ("top_45", (45, 0, 1, 0, 0)),
    ("left_-90", (0, -90, 1, 0, 0)), ("right_90", (0, 90, 1, 0, 0)),
)


def _sphere_poses(theta_deg: torch.Tensor, phi_deg: torch.Tensor, r: torch.Tensor, x: Optional[torch.Tensor] = None,
                  y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`sphere2pose` (models/utils.py:83-131) of the anchor camera c2w_init (demo.py:553-564) for all frames at once: translate along
    the world z axis (and x / y), then rotate about x by theta and about y by phi.  fp32 vectors [F] -> [F,4,4]."""
    n = theta_deg.shape[0]
    th, ph = torch.deg2rad(theta_deg), torch.deg2rad(phi_deg)
    c2w = torch.diag(torch.tensor([-1.0, 1.0, -1.0, 1.0])).repeat(n, 1, 1)
    c2w[:, 2, 3] -= r
    if x is not None:
        c2w[:, 1, 3] += y
        c2w[:, 0, 3] -= x
    one, zero = torch.ones(n), torch.zeros(n)
    rows = lambda *v: torch.stack(v, dim=-1)
    rot_x = torch.stack([rows(one, zero, zero, zero), rows(zero, th.cos(), -th.sin(), zero),
                         rows(zero, th.sin(), th.cos(), zero), rows(zero, zero, zero, one)], dim=1)
    rot_y = torch.stack([rows(ph.cos(), zero, ph.sin(), zero), rows(zero, one, zero, zero),
                         rows(-ph.sin(), zero, ph.cos(), zero), rows(zero, zero, zero, one)], dim=1)
    return rot_y @ (rot_x @ c2w)


def _key_interpolation(keys: Sequence[float], n: int) -> np.ndarray:
    """`generate_traj_txt`'s per-frame values from the key values of a trajectory file (models/utils.py:161-202): more than 3 keys ->
    cubic smoothing spline (scipy UnivariateSpline, default smoothing) with the two end points pinned, else piecewise linear."""
    from scipy.interpolate import UnivariateSpline, interp1d
    keys = [float(v) for v in keys]
    x, xn = np.linspace(0, 1, len(keys)), np.linspace(0, 1, n)
    if len(keys) > 3:
        v = UnivariateSpline(x, keys, k=3)(xn)
        v[0], v[-1] = keys[0], keys[-1]
        return v
    return interp1d(x, keys)(xn)


def traj_poses(depths: torch.Tensor, theta: Sequence[float], phi: Sequence[float], r: Sequence[float], num_frames: int,
               radius_scale: float = 1.0, anchor_idx: int = 0, device=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """`TrajCrafter.get_poses` with `opts.camera == 'traj'` (demo.py:566-573): the three lines of a trajectory file (test/trajs/*.txt:
    theta keys, phi keys, r keys in units of the orbit radius) interpolated over the clip (`generate_traj_txt`,
    models/utils.py:174-210) -> (pose_s, pose_t, K) like `orbit_poses`."""
    device = depths.device if device is None else torch.device(device)
    radius = min(float(depths[0, 0, depths.shape[-2] // 2, depths.shape[-1] // 2]) * radius_scale, 5.0)
    f32 = lambda v: torch.from_numpy(np.asarray(v).astype(np.float32))
    rs = _key_interpolation([float(v) * np.float32(radius) for v in r], num_frames)
    poses = _sphere_poses(f32(_key_interpolation(theta, num_frames)), f32(_key_interpolation(phi, num_frames)), f32(rs))
    poses[:, 2, 3] += np.float32(radius)                                                 # :580
    K = torch.tensor([[500.0, 0.0, 512.0], [0.0, 500.0, 288.0], [0.0, 0.0, 1.0]]).repeat(num_frames, 1, 1)
    pose_s = poses[anchor_idx:anchor_idx + 1].repeat(num_frames, 1, 1)
    return pose_s.to(device), poses.to(device), K.to(device)


def read_traj_txt(path: str):
    """A trajectory file of the reference (test/trajs/loop1.txt): line 1 theta keys, line 2 phi keys, line 3 r keys (demo.py:567-571)."""
    with open(path) as f:
        lines = f.readlines()
    return tuple([float(v) for v in lines[i].split()] for i in range(3))


def orbit_poses(depths: torch.Tensor, target_pose: Sequence[float], num_frames: int, radius_scale: float = 1.0,
                anchor_idx: int = 0, device=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """`TrajCrafter.get_poses` with `opts.camera == 'target'` (demo.py:538-586): the camera moves from the anchor pose to
    `target_pose = (d_theta, d_phi, d_r, d_x, d_y)` linearly over the clip on a sphere around the scene centre
    (`generate_traj_specified` / `sphere2pose`, models/utils.py:83-158).  All frames at once (the reference builds one 4x4
    per frame in a Python loop).  -> (pose_s [F,4,4] = the anchor repeated, pose_t [F,4,4], K [F,3,3]) in fp32 on `device`.
    The orbit radius is the depth at the centre pixel of frame 0, times `radius_scale`, clamped to 5."""
    device = depths.device if device is None else torch.device(device)
    radius = min(float(depths[0, 0, depths.shape[-2] // 2, depths.shape[-1] // 2]) * radius_scale, 5.0)
    d_theta, d_phi, d_r, d_x, d_y = (float(v) for v in target_pose)
    lin = lambda end: torch.from_numpy(np.linspace(0, end, num_frames).astype(np.float32))
    poses = _sphere_poses(lin(d_theta), lin(d_phi), lin(d_r * np.float32(radius)), lin(d_x), lin(d_y))
    poses[:, 2, 3] += np.float32(radius)                                                 # :580
    K = torch.tensor([[500.0, 0.0, 512.0], [0.0, 500.0, 288.0], [0.0, 0.0, 1.0]]).repeat(num_frames, 1, 1)   # :545-552
    pose_s = poses[anchor_idx:anchor_idx + 1].repeat(num_frames, 1, 1)
    return pose_s.to(device), poses.to(device), K.to(device)


def render_conditioning(warper: Warper, frames: torch.Tensor, depths: torch.Tensor, pose_s: torch.Tensor, pose_t: torch.Tensor,
                        K: torch.Tensor, sample_size: Tuple[int, int], mask: bool = False,
                        ref_frames: int = 10) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """demo.py:75-120.  frames [T,3,H,W] in [-1,1], depths [T,1,H,W], poses [T,4,4], K [T,3,3] ->
    (cond_video [1,3,T,h,w] in [0,1], cond_masks [1,1,T,h,w] in {0,255} with 255 = hole, frames_ref [1,3,ref_frames,h,w] in [0,1]):
    the `video`, `mask_video`, `reference` arguments of `TrajCrafter_Pipeline.__call__`."""
    dev = warper.device
    frames = frames.to(dev, torch.float32)
    warped, mask2, _, _ = warper.forward_warp(frames, None, depths, pose_s, pose_t, K, None, mask, twice=False, per_frame=True)
    cond_video = (warped + 1.0) / 2.0                                             # :91
    frames_r = F.interpolate(frames, size=sample_size, mode="bilinear", align_corners=False)        # :94-96
    cond_video = F.interpolate(cond_video, size=sample_size, mode="bilinear", align_corners=False)  # :97-99
    cond_masks = F.interpolate(mask2, size=sample_size, mode="nearest")                             # :100
    frames01 = (frames_r.permute(1, 0, 2, 3).unsqueeze(0) + 1.0) / 2.0            # :117
    return (cond_video.permute(1, 0, 2, 3).unsqueeze(0),                          # :119
            (1.0 - cond_masks.permute(1, 0, 2, 3).unsqueeze(0)) * 255.0,           # :120
            frames01[:, :, :ref_frames])                                           # :118


def render_and_generate(pipe, warper: Warper, frames: torch.Tensor, depths: torch.Tensor, pose_s: torch.Tensor,
                        pose_t: torch.Tensor, K: torch.Tensor, sample_size: Tuple[int, int] = (384, 672),
                        prompt: Optional[str] = None, negative_prompt: Optional[str] = None,
                        prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                        guidance_scale: float = 6.0, num_inference_steps: int = 50, seed: int = 43, mask: bool = False,
                        output_type: str = "numpy", **pipe_kwargs) -> torch.Tensor:
    """demo.py:75-148: render the clip into the target views, then denoise it.  Returns `.videos` of the pipeline."""
    video, mask_video, reference = render_conditioning(warper, frames, depths, pose_s, pose_t, K, sample_size, mask)
    gen = torch.Generator(device=warper.device).manual_seed(seed)                 # :121
    return pipe(prompt, num_frames=frames.shape[0], negative_prompt=negative_prompt, height=sample_size[0], width=sample_size[1],
                generator=gen, guidance_scale=guidance_scale, num_inference_steps=num_inference_steps, video=video,
                mask_video=mask_video, reference=reference, prompt_embeds=prompt_embeds,
                negative_prompt_embeds=negative_prompt_embeds, output_type=output_type, **pipe_kwargs).videos


def run_orbits(pipe, warper: Warper, frames: torch.Tensor, depths: torch.Tensor,
               variants: Sequence[Tuple[str, Sequence[float]]] = ORBIT_VARIANTS, radius: float = 1.0,
               radius_scale: float = 1.0, K: Optional[torch.Tensor] = None, sample_size: Tuple[int, int] = (384, 672),
               prompt: Optional[str] = None, negative_prompt: Optional[str] = None,
               prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
               guidance_scale: float = 6.0, num_inference_steps: int = 50, seed: int = 43, mask: bool = True,
               gather: bool = True, **pipe_kwargs) -> torch.Tensor:
    """BASELINE configs[3]: the trajectory variants of ONE clip, data-parallel (inference_orbits.py:248-300).

    The reference runs `variants` sequentially on one GPU (or as separate single-GPU SLURM array tasks): per variant the target
    poses (`get_poses`), the point-cloud render of the clip into those views with `opts.mask = True`, the three resizes, then
    `pipeline(...)` with a fresh `Generator(seed)` (demo.py:75-148).  The variants are independent, so here rank r of the
    `torch.distributed` group (one process per GPU, RCCL) owns variants r, r + W, ... — render, VAE encodes, the denoise loop and
    the decode all run on that rank's GPU with NO collective — and ONE all-gather of the decoder's bf16 channels-last output
    reassembles the clips; the fp32 frame conversion of `decode_latents` (:514-517) runs on the gathered tensor (bit-identical to
    converting per rank, half the bytes on the wire).  Without an initialised process group this is the reference's sequential loop.

    A variant is (name, (d_theta, d_phi, d_r, d_x, d_y)) — a target pose, `camera == 'target'` — or (name, (theta keys, phi keys,
    r keys)) — the three lines of a trajectory file, `camera == 'traj'` (demo.py:566-573; `read_traj_txt`).
    frames [F,3,H,W] in [-1,1] and depths [F,1,H,W] come from the conditioning stage (video decode + depth estimator: control
    plane, not on this path).  `K` overrides the reference's hard-wired 1024x576 intrinsics (demo.py:545-552) for other input
    sizes.  Every variant uses `Generator(seed)` like the reference (demo.py:121); the reference's one unseeded draw — the
    posterior sample of the reference-frame latents, global RNG (pipeline :885-889) — is pinned per variant
    (`torch.manual_seed(seed + 1 + i)`) so that the result does not depend on the number of ranks.
    -> fp32 frames [len(variants),3,F,h,w] in [0,1] on this rank's GPU, ordered like `variants`; `gather=False` -> only this
    rank's variants (rank order r, r + W, ...)."""
    names = [v[0] for v in variants]
    if len(set(names)) != len(names):
        raise ValueError(f"run_orbits: duplicate variant names {names}")
    n_frames = frames.shape[0]

    def run_one(i: int) -> torch.Tensor:
        spec = variants[i][1]
        if len(spec) == 3 and all(hasattr(v, "__len__") for v in spec):          # (theta keys, phi keys, r keys): `camera == 'traj'`
            theta, phi, r_keys = spec
            pose_s, pose_t, k_ref = traj_poses(depths, theta, phi, [float(v) * radius for v in r_keys], n_frames, radius_scale,
                                               device=warper.device)
        else:                                                                    # (d_theta, d_phi, d_r, d_x, d_y): `camera == 'target'`
            d_theta, d_phi, d_r, d_x, d_y = spec
            pose_s, pose_t, k_ref = orbit_poses(depths, (d_theta, d_phi, d_r * radius, d_x, d_y), n_frames, radius_scale,
                                                device=warper.device)
        video, mask_video, reference = render_conditioning(warper, frames, depths, pose_s, pose_t,
                                                           k_ref if K is None else K.to(warper.device), sample_size, mask)
        torch.manual_seed(seed + 1 + i)                                           # the posterior sample's global RNG, per variant
        gen = torch.Generator(device=warper.device).manual_seed(seed)             # demo.py:121
        return pipe(prompt, num_frames=n_frames, negative_prompt=negative_prompt, height=sample_size[0], width=sample_size[1],
                    generator=gen, guidance_scale=guidance_scale, num_inference_steps=num_inference_steps, video=video,
                    mask_video=mask_video, reference=reference, prompt_embeds=prompt_embeds,
                    negative_prompt_embeds=negative_prompt_embeds, output_type="cl_bf16", **pipe_kwargs).videos

    out_cl = dp.run_trajectories(run_one, len(variants), gather=gather)           # [n,F,h,w,3] bf16: the single collective
    return pipe.vae.cl_to_frames(out_cl)
