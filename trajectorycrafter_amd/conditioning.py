"""Conditioning hand-off: what the reference's CPU / other-GPU conditioning stage (depth estimation, point-cloud render,
captioning: demo.py:45-120) produces and `TrajCrafter_Pipeline.__call__` consumes (demo.py:121-148), as ONE `.safetensors`
file.  Tensor names are the reference's variable names at that seam:

    cond_video   [1,3,F,H,W] fp32 in [0,1]    -> pipeline kwarg `video`       (demo.py:91-99,119)
    cond_masks   [1,1,F,H,W] fp32 in {0,255}  -> pipeline kwarg `mask_video`  (demo.py:100,120; 255 = hole)
    frames_ref   [1,3,<=10,H,W] fp32 in [0,1] -> pipeline kwarg `reference`   (demo.py:117-118)
    prompt_embeds / negative_prompt_embeds [1,226,4096]  (T5 output, pipeline :831-843)  or the prompt strings in the metadata

optionally the pre-encoded form (skips the two VAE encodes; what benchmarks and the data-parallel runner ship to each rank):

    inpaint_latents [B|2B,T,17,h,w], ref_latents [B|2B,Tr,16,h,w], latents [B,T,16,h,w] (initial noise)

safetensors holds raw tensors + a string->string header: nothing is executed on load (unlike torch.save pickles).
"""
from __future__ import annotations

import json
from typing import Any, Dict, Optional

import torch

FORMAT = "trajectorycrafter-conditioning/1"
# file tensor name -> TrajCrafter_Pipeline.__call__ keyword
TENSOR_TO_KWARG = {
    "cond_video": "video", "cond_masks": "mask_video", "frames_ref": "reference",
    "prompt_embeds": "prompt_embeds", "negative_prompt_embeds": "negative_prompt_embeds",
    "inpaint_latents": "inpaint_latents", "ref_latents": "ref_latents", "latents": "latents",
}
_SCALARS = ("prompt", "negative_prompt", "height", "width", "num_frames", "seed", "guidance_scale", "num_inference_steps")


def save_conditioning(path: str, *, cond_video: Optional[torch.Tensor] = None, cond_masks: Optional[torch.Tensor] = None,
                      frames_ref: Optional[torch.Tensor] = None, prompt_embeds: Optional[torch.Tensor] = None,
                      negative_prompt_embeds: Optional[torch.Tensor] = None, inpaint_latents: Optional[torch.Tensor] = None,
                      ref_latents: Optional[torch.Tensor] = None, latents: Optional[torch.Tensor] = None,
                      **meta: Any) -> None:
    """Write the hand-off file.  Either the pixel form (cond_video + frames_ref [+ cond_masks]) or the pre-encoded form
    (inpaint_latents + ref_latents) must be complete; `meta` takes prompt / negative_prompt / height / width / num_frames / seed /
    guidance_scale / num_inference_steps (stored as JSON in the header)."""
    from safetensors.torch import save_file
    tensors = {k: v for k, v in dict(cond_video=cond_video, cond_masks=cond_masks, frames_ref=frames_ref,
                                     prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                                     inpaint_latents=inpaint_latents, ref_latents=ref_latents, latents=latents).items()
               if v is not None}
    pixel = "cond_video" in tensors and "frames_ref" in tensors
    encoded = "inpaint_latents" in tensors and "ref_latents" in tensors
    if not (pixel or encoded):
        raise ValueError("save_conditioning: give cond_video + frames_ref (+ cond_masks) or inpaint_latents + ref_latents")
    if pixel:
        v = tensors["cond_video"]
        if v.dim() != 5 or v.shape[1] != 3:
            raise ValueError(f"cond_video must be [B,3,F,H,W], got {tuple(v.shape)}")
        m = tensors.get("cond_masks")
        if m is not None and (m.dim() != 5 or m.shape[1] != 1 or m.shape[2:] != v.shape[2:]):
            raise ValueError(f"cond_masks must be [B,1,F,H,W] matching cond_video, got {tuple(m.shape)}")
    unknown = set(meta) - set(_SCALARS)
    if unknown:
        raise ValueError(f"save_conditioning: unknown metadata {sorted(unknown)}; allowed: {_SCALARS}")
    header = {"format": FORMAT, "meta": json.dumps({k: v for k, v in meta.items() if v is not None})}
    save_file({k: t.detach().contiguous().cpu() for k, t in tensors.items()}, path, metadata=header)


def load_conditioning(path: str, device: Optional[torch.device] = None) -> Dict[str, Any]:
    """-> keyword arguments for `TrajCrafter_Pipeline.__call__` (tensors renamed to the pipeline's names and moved to
    `device`; prompt / sizes / step count from the header when present; `seed` becomes a `generator` on `device`)."""
    from safetensors import safe_open
    kw: Dict[str, Any] = {}
    with safe_open(path, framework="pt") as f:
        header = f.metadata() or {}
        if header.get("format") != FORMAT:
            raise ValueError(f"{path}: not a {FORMAT} file (format = {header.get('format')!r})")
        for name in f.keys():
            if name not in TENSOR_TO_KWARG:
                raise ValueError(f"{path}: unknown tensor {name!r}")
            t = f.get_tensor(name)
            kw[TENSOR_TO_KWARG[name]] = t.to(device) if device is not None else t
    meta = json.loads(header.get("meta", "{}"))
    seed = meta.pop("seed", None)
    kw.update(meta)
    if seed is not None and "latents" not in kw:
        kw["generator"] = torch.Generator(device=device if device is not None else "cpu").manual_seed(int(seed))   # demo.py:121
    if "prompt_embeds" in kw:
        kw.pop("prompt", None), kw.pop("negative_prompt", None)       # the pipeline refuses both forms at once (:559-563)
    kw.setdefault("prompt", None)
    return kw
