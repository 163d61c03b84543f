"""Oracle for the orbit camera poses (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows the reference frame by frame: `sphere2pose` models/utils.py:83-131, `generate_traj_specified`
models/utils.py:134-158 (one pose per frame in a Python loop, angles cast to float32 per frame) and the
`camera == 'target'` branch of `TrajCrafter.get_poses` demo.py:538-586.  Pinned by
tests/golden/orbit_poses.safetensors (the reference's own `generate_traj_specified` on the eight variants of
inference_orbits.py:258-283, written by tests/golden/make_golden.py poses).
"""
from __future__ import annotations

import numpy as np
import torch


def sphere2pose(c2w: torch.Tensor, theta, phi, r, x=None, y=None) -> torch.Tensor:
    """reference models/utils.py:83-131.  c2w [1,4,4]; theta / phi in degrees (float32 scalars)."""
    c2w = c2w.clone()
    c2w[:, 2, 3] -= r
    if x is not None:
        c2w[:, 1, 3] += y
    if y is not None:
        c2w[:, 0, 3] -= x
    th = torch.deg2rad(torch.tensor(theta))
    s, c = torch.sin(th), torch.cos(th)
    rot_x = torch.tensor([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]]).unsqueeze(0)
    ph = torch.deg2rad(torch.tensor(phi))
    s, c = torch.sin(ph), torch.cos(ph)
    rot_y = torch.tensor([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]]).unsqueeze(0)
    return torch.matmul(rot_y, torch.matmul(rot_x, c2w))


def generate_traj_specified(c2w_anchor: torch.Tensor, theta, phi, d_r, d_x, d_y, frame: int) -> torch.Tensor:
    """reference models/utils.py:134-158 -> [frame,4,4]."""
    out = []
    lin = lambda end: np.linspace(0, float(end), frame)      # float(): d_r arrives as a 0-dim fp32 tensor from get_poses (same value)
    for th, ph, r, x, y in zip(lin(theta), lin(phi), lin(d_r), lin(d_x), lin(d_y)):
        out.append(sphere2pose(c2w_anchor, np.float32(th), np.float32(ph), np.float32(r), np.float32(x), np.float32(y)))
    return torch.cat(out, dim=0)


def get_poses_target(depths: torch.Tensor, target_pose, num_frames: int, radius_scale: float = 1.0, anchor_idx: int = 0):
    """reference demo.py:538-586, `opts.camera == 'target'` -> (pose_s, pose_t, K), each [num_frames, ...]."""
    radius = depths[0, 0, depths.shape[-2] // 2, depths.shape[-1] // 2].cpu() * radius_scale
    radius = min(radius, 5)
    K = torch.tensor([[500, 0.0, 512.0], [0.0, 500, 288.0], [0.0, 0.0, 1.0]]).repeat(num_frames, 1, 1)
    c2w_init = torch.tensor([[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]).unsqueeze(0)
    dtheta, dphi, dr, dx, dy = target_pose
    poses = generate_traj_specified(c2w_init, dtheta, dphi, dr * radius, dx, dy, num_frames)
    poses[:, 2, 3] = poses[:, 2, 3] + radius
    pose_s = poses[anchor_idx:anchor_idx + 1].repeat(num_frames, 1, 1)
    return pose_s, poses, K


def txt_interpolation(input_list, n: int, mode: str = "smooth"):
    """reference models/utils.py:161-171: scipy UnivariateSpline (k = 3, default smoothing) or interp1d over [0, 1]."""
    from scipy.interpolate import UnivariateSpline, interp1d
    x = np.linspace(0, 1, len(input_list))
    if mode == "smooth":
        f = UnivariateSpline(x, input_list, k=3)
    elif mode == "linear":
        f = interp1d(x, input_list)
    else:
        raise KeyError(f"Invalid txt interpolation mode: {mode}")
    return f(np.linspace(0, 1, n))


def generate_traj_txt(c2w_anchor: torch.Tensor, phi, theta, r, frame: int) -> torch.Tensor:
    """reference models/utils.py:174-210: per-frame (theta, phi, r) from the key values of a trajectory file — smoothing spline with
    the end points pinned for more than 3 keys, linear otherwise — each through `sphere2pose` -> [frame,4,4]."""
    def interp(keys):
        if len(keys) > 3:
            v = txt_interpolation(keys, frame, mode="smooth")
            v[0], v[-1] = keys[0], keys[-1]
            return v
        return txt_interpolation(keys, frame, mode="linear")
    phis, thetas, rs = interp(phi), interp(theta), interp(r)
    return torch.cat([sphere2pose(c2w_anchor, np.float32(th), np.float32(ph), np.float32(rr)) for th, ph, rr in zip(thetas, phis, rs)], dim=0)


def get_poses_traj(depths: torch.Tensor, theta, phi, r, num_frames: int, radius_scale: float = 1.0, anchor_idx: int = 0):
    """reference demo.py:538-586, `opts.camera == 'traj'`: the three lines of a trajectory file (theta keys, phi keys, r keys; r in
    units of the orbit radius) -> (pose_s, pose_t, K)."""
    radius = depths[0, 0, depths.shape[-2] // 2, depths.shape[-1] // 2].cpu() * radius_scale
    radius = min(radius, 5)
    K = torch.tensor([[500, 0.0, 512.0], [0.0, 500, 288.0], [0.0, 0.0, 1.0]]).repeat(num_frames, 1, 1)
    c2w_init = torch.tensor([[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]).unsqueeze(0)
    poses = generate_traj_txt(c2w_init, list(phi), list(theta), [float(i) * radius for i in r], num_frames)
    poses[:, 2, 3] = poses[:, 2, 3] + radius
    pose_s = poses[anchor_idx:anchor_idx + 1].repeat(num_frames, 1, 1)
    return pose_s, poses, K
