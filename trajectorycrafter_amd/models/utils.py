"""MI355X-native `Warper` — the point-cloud render of reference models/utils.py:213-293.

Built: `forward_warp(..., mask=False|True, twice=False)` — what demo.py calls (`--mask` selects `clean_points`, the
5x5 dilation of the holes, done here inside the resolve kernel instead of a cv2 round trip through the host; with
mask=True the reference returns float64 because of a numpy promotion, this returns fp32).  The 4x4 / 3x3 inverses are tiny host-side
torch ops; projection, splatting (float atomics) and normalisation are HIP kernels (`tcx_warp_forward`).
`twice=True` (reference :294-347, mask=False): that fused stage, then the flow, the warped frame and the warped depth through the
generic splat `tcx_bilinear_splat`."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import ops
from .._lib import TcxError


class Warper:
    def __init__(self, resolution: tuple = None, device: str = "cuda:0"):
        self.resolution = resolution
        self.device = torch.device("cuda:0" if device in ("gpu0", "cuda") else device)
        self.dtype = torch.float32

    def forward_warp(self, frame1: torch.Tensor, mask1: Optional[torch.Tensor], depth1: torch.Tensor,
                     transformation1: torch.Tensor, transformation2: torch.Tensor, intrinsic1: torch.Tensor,
                     intrinsic2: Optional[torch.Tensor], mask=False, twice=False, per_frame: bool = False
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """reference :220-293 -> (warped_frame2 [b,3,h,w] in [-1,1], mask2 [b,1,h,w], warped_depth2 [b,1,h,w], flow12 [b,2,h,w]).
        per_frame=True (not in the reference): the b items are b independent batch-1 reference calls — the whole clip
        of demo.py:100-116's per-frame loop in one launch."""
        if twice and (mask or per_frame):
            raise NotImplementedError("forward_warp(twice=True) is built for mask=False (its only caller, notebooks/15_10_25_depth/"
                                      "collect_dataset.py:272-282; with mask=True the reference dilates an already dilated mask, :307-343)")
        if self.device.type != "cuda":
            raise TcxError("Warper: the HIP splat needs a GPU device (no CPU fallback; use oracle.warp on the CPU)")
        if self.resolution is not None:
            assert tuple(frame1.shape[2:4]) == tuple(self.resolution)
        b, c, h, w = frame1.shape
        assert frame1.shape == (b, 3, h, w) and depth1.shape == (b, 1, h, w)
        assert transformation1.shape == (b, 4, 4) and transformation2.shape == (b, 4, 4) and intrinsic1.shape == (b, 3, 3)
        if mask1 is not None:
            assert mask1.shape == (b, 1, h, w)
        if intrinsic2 is None:
            intrinsic2 = intrinsic1
        to = dict(device=self.device, dtype=self.dtype)
        t1, t2, k1, k2 = (t.to(**to) for t in (transformation1, transformation2, intrinsic1, intrinsic2))
        rel = torch.bmm(t2, torch.linalg.inv(t1))                                 # :365-367
        mats = torch.cat([torch.linalg.inv(k1).reshape(b, 9), rel[:, :3, :].reshape(b, 12), k2.reshape(b, 9)], dim=1).contiguous()
        m1 = None if mask1 is None else mask1.to(**to).contiguous()
        if not twice:
            return ops.warp_forward(frame1.to(**to).contiguous(), m1, depth1.to(**to).contiguous(), mats, per_item_max=per_frame,
                                    clean_points=bool(mask))
        # reference :294-347: warp to the target view, splat the flow itself the same way, then splat frame and depth BACK along the
        # negated warped flow with the warped depth as the weight (the target-view occlusions end up as holes in the source view)
        warped2, mask2, wdepth2, flow12, tdepth = ops.warp_forward(frame1.to(**to).contiguous(), m1, depth1.to(**to).contiguous(), mats,
                                                                   return_tdepth=True)
        warped_flow, _ = ops.bilinear_splat(flow12, m1, tdepth, flow12, is_image=False)
        d2 = wdepth2[:, 0].contiguous()
        twice_frame1, twice_mask1 = ops.bilinear_splat(warped2, mask2, d2, warped_flow, is_image=True, flow_scale=-1.0)
        twice_depth1, _ = ops.bilinear_splat(wdepth2, mask2, d2, warped_flow, is_image=False, flow_scale=-1.0)
        return twice_frame1, twice_mask1, twice_depth1, None

    # ---- the reference's other public helpers that have a place on this path ----
    def bilinear_splatting(self, frame1: torch.Tensor, mask1: Optional[torch.Tensor], depth1: torch.Tensor, flow12: torch.Tensor,
                           flow12_mask: Optional[torch.Tensor] = None, is_image: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """reference :422-583 on `tcx_bilinear_splat`: frame1 [b,c<=4,h,w], depth1 [b,h,w] (the reference's callers pass
        `trans_depth1`, :277-291), flow12 [b,2,h,w] -> (warped_frame2, mask2).  `flow12_mask` multiplies the weights like mask1 (:497-515)."""
        if self.device.type != "cuda":
            raise TcxError("Warper: the HIP splat needs a GPU device (no CPU fallback; use oracle.warp on the CPU)")
        to = dict(device=self.device, dtype=self.dtype)
        m = None if mask1 is None else mask1.to(**to)
        if flow12_mask is not None:
            m = flow12_mask.to(**to) if m is None else m * flow12_mask.to(**to)
        d = depth1.to(**to)
        if d.dim() == 4:
            d = d[:, 0]
        return ops.bilinear_splat(frame1.to(**to).contiguous(), None if m is None else m.contiguous(), d.contiguous(),
                                  flow12.to(**to).contiguous(), is_image=bool(is_image))

    @staticmethod
    def create_grid(b: int, h: int, w: int) -> torch.Tensor:
        """reference :628-636: [b, 2, h, w] of (x, y) pixel coordinates (int64, CPU)."""
        ys, xs = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
        return torch.stack([xs, ys], dim=0)[None].repeat([b, 1, 1, 1])

    @staticmethod
    def camera_intrinsic_transform(capture_width=1920, capture_height=1080, patch_start_point: tuple = (0, 0)):
        """reference :656-666: the 4x4 intrinsic matrix (focal 2100) of a crop starting at patch_start_point = (y, x)."""
        import numpy
        start_y, start_x = patch_start_point
        k = numpy.eye(4)
        k[0, 0] = k[1, 1] = 2100
        k[0, 2] = capture_width / 2.0 - start_x
        k[1, 2] = capture_height / 2.0 - start_y
        return k

    @staticmethod
    def get_device(device: str):
        """reference :668-682: 'cpu' | 'gpuN' -> torch.device."""
        if device.startswith("gpu") and torch.cuda.is_available():
            return torch.device(f"cuda:{int(device[3:])}")
        return torch.device("cpu")

