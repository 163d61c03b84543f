"""Thin driver: point-cloud render -> conditioning -> denoise -> frames, all on one MI355X.

Mirrors the data flow of the reference's `TrajCrafter.infer_*` (demo.py:75-148) from the point where frames, depths,
camera poses and the prompt embedding exist: those come from the reference's depth estimator, pose generator and
captioner, which are control plane / other models and out of scope.  Everything between is on the GPU:
`Warper.forward_warp` (one launch for the clip), the three resizes (torch ops on device memory), the HIP VAE encoder inside
`TrajCrafter_Pipeline.__call__`, the denoise loop and the decode.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .models.utils import Warper


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
