"""Oracle for `TrajCrafter_Pipeline.__call__` (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference models/pipeline_trajectorycrafter.py: get_resize_crop_region_for_grid :43-58,
resize_mask :127-160, prepare_latents :383-457, prepare_mask_latents :459-506, decode_latents
:508-518, _prepare_rotary_positional_embeddings :616-649, __call__ :673-1216.  Prompt embeddings
are passed in (the T5 encoder is conditioning I/O, out of scope).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F

from . import diffusers_restated as dr
from . import transformer as otr
from . import vae as ovae
from .prec import Prec


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """reference :43-58."""
    tw, th = tgt_width, tgt_height
    h, w = src
    r = h / w
    if r > (th / tw):
        resize_height = th
        resize_width = int(round(th / h * w))
    else:
        resize_width = tw
        resize_height = int(round(tw / w * h))
    crop_top = int(round((th - resize_height) / 2.0))
    crop_left = int(round((tw - resize_width) / 2.0))
    return (crop_top, crop_left), (crop_top + resize_height, crop_left + resize_width)


def prepare_rotary(height: int, width: int, num_latent_frames: int, patch: int, head_dim: int, vae_sf: int = 8):
    """reference :616-649."""
    gh, gw = height // (vae_sf * patch), width // (vae_sf * patch)
    base_w, base_h = 720 // (vae_sf * patch), 480 // (vae_sf * patch)
    crops = get_resize_crop_region_for_grid((gh, gw), base_w, base_h)
    return dr.get_3d_rotary_pos_embed(head_dim, crops, (gh, gw), num_latent_frames)


def resize_mask(mask: torch.Tensor, latent: torch.Tensor) -> torch.Tensor:
    """reference :127-160 (process_first_frame_only=True).  mask [B,1,F,H,W], latent [B,C,T,h,w]."""
    tgt = list(latent.shape[2:])
    first = F.interpolate(mask[:, :, 0:1], size=[1] + tgt[1:], mode="trilinear", align_corners=False)
    if tgt[0] - 1 != 0:
        rest = F.interpolate(mask[:, :, 1:], size=[tgt[0] - 1] + tgt[1:], mode="trilinear", align_corners=False)
        return torch.cat([first, rest], dim=2)
    return first


def build_conditioning(vae_sd, vae_cfg, video, mask_video, reference, height, width, prec="fp32",
                       ref_generator: Optional[torch.Generator] = None, do_cfg: bool = True):
    """reference :862-897 and :927-1028 -> (inpaint_latents [2B,T,17,h,w], ref_input [2B,Tr,16,h,w])."""
    p = Prec(prec)
    sf = ovae.DEFAULT_CONFIG["scaling_factor"] if "scaling_factor" not in vae_cfg else vae_cfg["scaling_factor"]
    B, _, Fv = video.shape[:3]

    def prep(x, **kw):
        b, c, f = x.shape[:3]
        y = dr.vae_image_preprocess(x.permute(0, 2, 1, 3, 4).reshape(b * f, c, *x.shape[3:]).float(), height, width, **kw)
        return y.reshape(b, f, c, height, width).permute(0, 2, 1, 3, 4)

    init_video = prep(video)
    ref_video = prep(reference)
    ref_lat = ovae.vae_encode(vae_sd, vae_cfg, ref_video, prec).sample(ref_generator) * sf       # :885-889
    ref_lat = p.R(ref_lat).permute(0, 2, 1, 3, 4)
    T = (Fv - 1) // 4 + 1
    h, w = height // 8, width // 8
    if bool((mask_video == 255).all()):                                                       # :928-948
        mask_lat = torch.zeros(B, T, 1, h, w)
        mv_lat = torch.zeros(B, T, 16, h, w)
    else:
        mask_cond = prep(mask_video, do_normalize=False, do_binarize=True)                    # :952-960
        tile = mask_cond.repeat(1, 3, 1, 1, 1)
        masked_video = init_video * (tile < 0.5) + torch.ones_like(init_video) * (tile > 0.5) * -1   # :969-974
        mv = ovae.vae_encode(vae_sd, vae_cfg, masked_video, prec).mode() * sf                 # :498-502
        mv = p.R(mv)
        mask_lat = resize_mask(1 - mask_cond, mv) * sf                                        # :991-996
        mask_lat = mask_lat.permute(0, 2, 1, 3, 4)
        mv_lat = mv.permute(0, 2, 1, 3, 4)
    rep = 2 if do_cfg else 1
    inpaint = p.R(torch.cat([torch.cat([mask_lat] * rep), torch.cat([mv_lat] * rep)], dim=2))   # :1026-1028
    return p.out(inpaint), p.out(torch.cat([ref_lat] * rep))


def denoise(tr_sd, tr_cfg, latents, prompt_embeds, negative_prompt_embeds, inpaint_latents, ref_input,
            height, width, num_inference_steps=50, guidance_scale=6.0, prec="fp32",
            scheduler: Optional[dr.DDIMScheduler] = None, num_blocks: Optional[int] = None,
            on_step: Optional[Callable] = None, strength: float = 1.0, video_latents: Optional[torch.Tensor] = None,
            step_noise: Optional[Callable] = None):
    """reference :1076-1198: the 50-step loop.  `latents` [B,T,16,h,w]; returns latents (activation dtype).
    strength < 1 (:664-671, :431-436): the loop starts `int(steps * strength)` steps before the end; when `video_latents`
    is given, `latents` is the NOISE and the start point is scheduler.add_noise(video_latents, noise, first timestep).
    `scheduler`: any of the restated samplers of demo.py:647-654; the model input goes through `scale_model_input` (:1099-1101);
    `step_noise(i, shape)` supplies the fp32 noise "Euler A" draws per step (the caller owns the generator semantics)."""
    p = Prec(prec)
    sched = scheduler or dr.DDIMScheduler()
    do_cfg = guidance_scale > 1.0
    pe = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0) if do_cfg else prompt_embeds
    sched.set_timesteps(num_inference_steps)
    cfg = dict(otr.DEFAULT_CONFIG)
    cfg.update(tr_cfg)
    rotary = prepare_rotary(height, width, latents.shape[1], cfg["patch_size"], cfg["attention_head_dim"])
    init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
    timesteps = sched.timesteps[max(num_inference_steps - init_timestep, 0):]
    if strength < 1.0 and video_latents is not None:
        lat = p.R(sched.add_noise(p, video_latents, latents, int(timesteps[0])))
    else:
        lat = p.R(latents * float(sched.init_noise_sigma))
    for i, t in enumerate(timesteps):
        x = torch.cat([lat] * 2) if do_cfg else lat
        x = sched.scale_model_input(p, x, t)                                                        # :1099-1101
        ts = t.expand(x.shape[0])
        noise_pred = otr.transformer_forward(tr_sd, cfg, x, pe, ts, inpaint_latents.float(), ref_input.float(),
                                             rotary, prec=prec, num_blocks=num_blocks).float()      # :1108-1117
        if do_cfg:
            u, c = noise_pred.chunk(2)
            noise_pred = u + guidance_scale * (c - u)                                               # :1157-1161
        if getattr(sched, "ancestral", False):
            lat = p.R(sched.step(p, noise_pred, t, p.out(lat), noise=step_noise(i, lat.shape)))
        else:
            lat = p.R(sched.step(p, noise_pred, t if isinstance(sched, dr._SigmaScheduler) else int(t), p.out(lat)))   # :1164-1178
        if on_step is not None:
            on_step(i, int(t), lat)
    return p.out(lat)


def decode_latents(vae_sd, vae_cfg, latents, prec="fp32"):
    """reference :508-518 -> float32 frames [B,3,F,H,W] in [0,1]."""
    p = Prec(prec)
    sf = vae_cfg.get("scaling_factor", ovae.DEFAULT_CONFIG["scaling_factor"])
    z = p.R(1 / sf * latents.float().permute(0, 2, 1, 3, 4))
    frames = ovae.vae_decode(vae_sd, vae_cfg, z, prec).float()
    return p.R(frames / 2 + 0.5).clamp(0, 1).float()


def pipeline_call(tr_sd, tr_cfg, vae_sd, vae_cfg, *, prompt_embeds, negative_prompt_embeds, video, mask_video,
                  reference, height, width, latents, num_inference_steps=50, guidance_scale=6.0, prec="fp32",
                  ref_generator=None, num_frames=49, output_type="numpy"):
    """reference `__call__` :673-1216 with prompt embeddings and initial latents supplied."""
    if num_frames > 49:
        raise ValueError("The number of frames must be less than 49 for now due to static positional embeddings.")
    if height % 8 != 0 or width % 8 != 0:
        raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
    do_cfg = guidance_scale > 1.0
    inpaint, ref_input = build_conditioning(vae_sd, vae_cfg, video, mask_video, reference, height, width, prec,
                                            ref_generator, do_cfg)
    lat = denoise(tr_sd, tr_cfg, latents, prompt_embeds, negative_prompt_embeds, inpaint, ref_input, height, width,
                  num_inference_steps, guidance_scale, prec)
    if output_type == "latent":
        return lat
    return decode_latents(vae_sd, vae_cfg, lat, prec)
