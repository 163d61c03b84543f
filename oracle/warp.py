"""Oracle for the point-cloud render `Warper.forward_warp(mask=False, twice=False)` — SURVEY §8(f) row f3
(TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference models/utils.py: compute_transformed_points :350-421, bilinear_splatting :422-583,
create_grid :626-634, forward_warp :220-293 (the `twice=False` branch; `clean_points` :585-626 for mask=True).  Pure torch, fp32, CPU.  Pinned by tests/golden/warp_tiny.safetensors
(generated from the reference's own Warper).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def transformed_points(depth: torch.Tensor, t1: torch.Tensor, t2: torch.Tensor, k1: torch.Tensor,
                       k2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reference :350-421 -> K2 [R|t] (depth * K1^-1 (x, y, 1)), [b, h, w, 3]; points with z <= 0.01 in the target
    camera are replaced by (1000, 1000, 1000)."""
    b, _, h, w = depth.shape
    k2 = k1 if k2 is None else k2
    rel = torch.bmm(t2, torch.linalg.inv(t1))                                  # [b,4,4]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=depth.dtype), torch.arange(w, dtype=depth.dtype), indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1)                  # [h,w,3]
    rays = torch.einsum("bij,hwj->bhwi", torch.linalg.inv(k1), pix)           # K1^-1 (x,y,1)
    world = depth[:, 0, :, :, None] * rays                                     # [b,h,w,3]
    cam2 = torch.einsum("bij,bhwj->bhwi", rel[:, :3, :3], world) + rel[:, None, None, :3, 3]
    proj = torch.einsum("bij,bhwj->bhwi", k2, cam2)
    behind = cam2[..., 2:3] <= 0.01
    return torch.where(behind.expand_as(proj), torch.full_like(proj, 1000.0), proj)


def bilinear_splat(frame: torch.Tensor, mask1: Optional[torch.Tensor], depth: torch.Tensor, flow: torch.Tensor,
                   is_image: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference :422-583.  frame [b,c,h,w], depth [b,h,w] (target-view depth of every source pixel), flow [b,2,h,w]."""
    b, c, h, w = frame.shape
    if mask1 is None:
        mask1 = torch.ones(b, 1, h, w, dtype=frame.dtype)
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack([xs, ys], 0)[None].to(frame)
    pos = flow + grid + 1                                                      # +1: one-pixel border of the accumulator
    fl, ce = torch.floor(pos).long(), torch.ceil(pos).long()
    lim = torch.tensor([w + 1, h + 1]).view(1, 2, 1, 1)
    pos = torch.minimum(torch.clamp(pos, min=0), lim.to(pos))
    fl = torch.minimum(torch.clamp(fl, min=0), lim)
    ce = torch.minimum(torch.clamp(ce, min=0), lim)
    fx, fy = pos[:, 0] - fl[:, 0], pos[:, 1] - fl[:, 1]                        # distance to the floor corner
    cx, cy = ce[:, 0] - pos[:, 0], ce[:, 1] - pos[:, 1]                        # distance to the ceil corner
    logd = torch.log(1 + torch.clamp(depth, min=0, max=1000))
    dw = torch.exp(logd / logd.max() * 50)                                     # nearer points win
    base = mask1[:, 0] / dw                                                    # [b,h,w]
    corners = (((fl[:, 1], fl[:, 0]), (1 - fy) * (1 - fx)), ((ce[:, 1], fl[:, 0]), (1 - cy) * (1 - fx)),
               ((fl[:, 1], ce[:, 0]), (1 - fy) * (1 - cx)), ((ce[:, 1], ce[:, 0]), (1 - cy) * (1 - cx)))
    acc = torch.zeros(b, (h + 2) * (w + 2), c, dtype=torch.float32)
    wsum = torch.zeros(b, (h + 2) * (w + 2), dtype=torch.float32)
    src = frame.permute(0, 2, 3, 1).reshape(b, h * w, c)
    for (iy, ix), pw in corners:
        wgt = (pw * base).reshape(b, h * w)
        idx = (iy * (w + 2) + ix).reshape(b, h * w)
        for n in range(b):
            acc[n].index_add_(0, idx[n], src[n] * wgt[n, :, None])
            wsum[n].index_add_(0, idx[n], wgt[n])
    acc = acc.view(b, h + 2, w + 2, c)[:, 1:-1, 1:-1].permute(0, 3, 1, 2)
    wsum = wsum.view(b, 1, h + 2, w + 2)[:, :, 1:-1, 1:-1]
    hit = wsum > 0
    out = torch.where(hit, acc / wsum, torch.tensor(-1.0 if is_image else 0.0))
    if is_image:
        out = torch.clamp(out, -1, 1)
    return out, hit.to(frame.dtype)


def clean_points(warped: torch.Tensor, mask2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference :585-626: holes (1 - mask2) dilated with cv2.dilate(ones(5,5)) — a 5x5 max filter whose border pixels
    are ignored (cv2's default border value for dilate) — then frame01 * (1 - holes) back to [-1,1].  cv2 is absent
    from this image: the dilation is restated from its documented semantics (parity unpinned for this sub-step); the
    reference's float64 promotion (numpy /255.0) is kept."""
    holes = (1 - mask2 >= 0.5).to(torch.float64)
    holes = torch.nn.functional.max_pool2d(holes, 5, stride=1, padding=2)      # pads with -inf: border ignored
    out = ((warped + 1.0) / 2.0) * (1 - holes)
    return out * 2.0 - 1.0, 1 - holes[:, 0:1]


def forward_warp(frame1, mask1, depth1, t1, t2, k1, k2=None, mask=False):
    """reference :220-293 (twice=False) -> (warped_frame2, mask2, warped_depth2, flow12)."""
    pts = transformed_points(depth1, t1, t2, k1, k2)
    coords = pts[..., :2] / pts[..., 2:3]
    tdepth = pts[..., 2]
    b, _, h, w = frame1.shape
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    flow = coords.permute(0, 3, 1, 2) - torch.stack([xs, ys], 0)[None].to(coords)
    warped, mask2 = bilinear_splat(frame1, mask1, tdepth, flow, True)
    wdepth, _ = bilinear_splat(tdepth[:, None], mask1, tdepth, flow, False)
    if mask:
        warped, mask2 = clean_points(warped, mask2)
    return warped, mask2, wdepth, flow


def forward_warp_twice(frame1, mask1, depth1, t1, t2, k1, k2=None):
    """reference :294-347 (`twice=True`, mask=False): the first warp, the flow splatted like an image, then frame and depth splatted
    back along the negated warped flow with the warped depth as weight -> (twice_warped_frame1, twice_warped_mask1,
    twice_warped_depth1, None)."""
    pts = transformed_points(depth1, t1, t2, k1, k2)
    coords = pts[..., :2] / pts[..., 2:3]
    tdepth = pts[..., 2]
    b, _, h, w = frame1.shape
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    flow = coords.permute(0, 3, 1, 2) - torch.stack([xs, ys], 0)[None].to(coords)
    warped2, mask2 = bilinear_splat(frame1, mask1, tdepth, flow, True)
    wdepth2, _ = bilinear_splat(tdepth[:, None], mask1, tdepth, flow, False)
    warped_flow, _ = bilinear_splat(flow, mask1, tdepth, flow, False)
    tw_frame, tw_mask = bilinear_splat(warped2, mask2, wdepth2[:, 0], -warped_flow, True)
    tw_depth, _ = bilinear_splat(wdepth2, mask2, wdepth2[:, 0], -warped_flow, False)
    return tw_frame, tw_mask, tw_depth, None
