"""MI355X-native `TrajCrafter_Pipeline` — drop-in for reference models/pipeline_trajectorycrafter.py.

Same constructor (:204-246) and `__call__` signature (:674-709); the 50-step loop (:1089-1198) runs
the HIP-backed `CrossTransformer3DModel`, the fused CFG + DDIM kernel and the HIP VAE decoder.
Nothing on this path syncs with the host inside the loop (timesteps / alphas are host scalars).

Two extension kwargs (ignored by reference callers): `inpaint_latents=` and `ref_latents=` take
pre-encoded conditioning (what :875-897 / :927-1028 build through `vae.encode`) so that benchmarks and
data-parallel runs can skip the pixel-space stage; without them the conditioning is built from
`video` / `mask_video` / `reference` with the HIP VAE encoder exactly as the reference does.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from .._lib import TcxError
from ..scheduler import DDIMScheduler

BF16 = torch.bfloat16


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


def get_3d_rotary_pos_embed(embed_dim: int, crops_coords, grid_size, temporal_size: int, theta: float = 10000.0):
    """diffusers get_3d_rotary_pos_embed(use_real=True): (cos, sin) fp32 [T*H*W, embed_dim], built on the host."""
    start, stop = crops_coords
    gh, gw = grid_size
    grid_h = np.linspace(start[0], stop[0], gh, endpoint=False, dtype=np.float32)
    grid_w = np.linspace(start[1], stop[1], gw, endpoint=False, dtype=np.float32)
    grid_t = np.linspace(0, temporal_size, temporal_size, endpoint=False, dtype=np.float32)
    dim_t, dim_h, dim_w = embed_dim // 4, embed_dim // 8 * 3, embed_dim // 8 * 3

    def axis(pos, dim):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        return torch.outer(torch.from_numpy(pos).float(), freqs).repeat_interleave(2, dim=-1)

    ft, fh, fw = axis(grid_t, dim_t), axis(grid_h, dim_h), axis(grid_w, dim_w)
    T = temporal_size
    freqs = torch.cat([ft[:, None, None, :].expand(T, gh, gw, dim_t), fh[None, :, None, :].expand(T, gh, gw, dim_h),
                       fw[None, None, :, :].expand(T, gh, gw, dim_w)], dim=-1).reshape(T * gh * gw, -1)
    return freqs.cos().contiguous(), freqs.sin().contiguous()


def retrieve_timesteps(scheduler, num_inference_steps: Optional[int] = None, device=None, timesteps: Optional[List[int]] = None,
                       sigmas: Optional[List[float]] = None, **kwargs):
    """reference :62-124 (a diffusers helper the reference module carries but its `__call__` never uses, :846): let the scheduler
    build its schedule — from a step count, or from caller-given `timesteps` / `sigmas` when its `set_timesteps` takes them — and
    return `(scheduler.timesteps, number of steps)`."""
    import inspect
    if timesteps is not None and sigmas is not None:
        raise ValueError("Only one of `timesteps` or `sigmas` can be passed. Please choose one to set custom values")
    accepted = set(inspect.signature(scheduler.set_timesteps).parameters)
    for name, custom in (("timesteps", timesteps), ("sigmas", sigmas)):
        if custom is None:
            continue
        if name not in accepted:
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support custom {name} "
                             "schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(device=device, **{name: custom}, **kwargs)
        return scheduler.timesteps, len(scheduler.timesteps)
    scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, num_inference_steps


def resize_mask(mask, latent, process_first_frame_only=True):
    """reference :127-160."""
    latent_size = latent.size()
    if process_first_frame_only:
        target_size = list(latent_size[2:])
        target_size[0] = 1
        first = F.interpolate(mask[:, :, 0:1], size=target_size, mode="trilinear", align_corners=False)
        target_size = list(latent_size[2:])
        target_size[0] = target_size[0] - 1
        if target_size[0] != 0:
            rest = F.interpolate(mask[:, :, 1:], size=target_size, mode="trilinear", align_corners=False)
            return torch.cat([first, rest], dim=2)
        return first
    return F.interpolate(mask, size=list(latent_size[2:]), mode="trilinear", align_corners=False)


def add_noise_to_reference_video(image: torch.Tensor, ratio: Optional[float] = None) -> torch.Tensor:
    """reference :163-175: per-sample Gaussian pixel noise (sigma = `ratio`, or exp(N(-3, 0.5)) when None), none on the
    masked-out pixels (== -1); drawn from the global RNG in `image`'s dtype on `image`'s device, as the reference does."""
    if ratio is None:
        sigma = torch.exp(torch.normal(mean=-3.0, std=0.5, size=(image.shape[0],)).to(image.device)).to(image.dtype)
    else:
        sigma = torch.ones((image.shape[0],)).to(image.device, image.dtype) * ratio
    noise = torch.randn_like(image) * sigma[:, None, None, None, None]
    noise = torch.where(image == -1, torch.zeros_like(image), noise)
    return image + noise


@dataclass
class CogVideoX_Fun_PipelineOutput:
    """reference :178-190."""
    videos: torch.Tensor


@dataclass
class DenoiseState:
    """What the reference's loop (:1089-1198) carries from iteration to iteration, plus its constant inputs."""
    latents: torch.Tensor                    # [B,T,16,h,w] bf16
    prompt_embeds: torch.Tensor              # [2B,226,4096] (negative first) when do_cfg
    inpaint_latents: torch.Tensor            # [2B,T,17,h,w]
    ref_input: torch.Tensor                  # [2B,Tr,16,h,w]
    image_rotary_emb: Optional[Tuple[torch.Tensor, torch.Tensor]]
    timesteps: List[int]
    num_inference_steps: int
    batch_size: int
    do_cfg: bool
    guidance_scale: float
    use_dynamic_cfg: bool = False
    generator: Optional[torch.Generator] = None   # the call's generator: "Euler A" / eta > 0 draw their per-step noise from it (:1073, :1166)
    eta: float = 0.0                              # handed to schedulers whose `step` takes it (DDIM), ignored by the others (:521-540)


class TrajCrafter_Pipeline:
    """reference :193-1216."""

    _optional_components: List[str] = []
    model_cpu_offload_seq = "text_encoder->transformer->vae"
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]

    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None):
        self.tokenizer, self.text_encoder = tokenizer, text_encoder
        self.vae, self.transformer = vae, transformer
        self.scheduler = scheduler if scheduler is not None else DDIMScheduler()
        if not hasattr(self.scheduler, "fused_cfg_step"):
            raise NotImplementedError(
                f"scheduler {type(self.scheduler).__name__} is not built on this path: the denoise loop fuses CFG + step into one "
                "kernel per scheduler class; built (trajectorycrafter_amd.scheduler): DDIMScheduler ('DDIM_Origin'), CogVideoXDDIMScheduler "
                "('DDIM_Cog'), EulerDiscreteScheduler ('Euler'), EulerAncestralDiscreteScheduler ('Euler A'), DPMSolverMultistepScheduler ('DPM++'), "
                "PNDMScheduler ('PNDM')")
        self.vae_scale_factor_spatial = 2 ** (len(self.vae.config.block_out_channels) - 1) if vae is not None else 8
        self.vae_scale_factor_temporal = int(self.vae.config.temporal_compression_ratio) if vae is not None else 4
        self.vae_scale_factor = self.vae_scale_factor_spatial
        self._guidance_scale, self._num_timesteps, self._interrupt = 6.0, 0, False
        self.last_timings: Dict[str, float] = {}

    # ---- plumbing the reference gets from DiffusionPipeline ----
    @classmethod
    def from_pretrained(cls, model_dir: str, vae=None, text_encoder=None, transformer=None, scheduler=None,
                        tokenizer=None, torch_dtype=None, **kw):
        """demo.py:659-666: components are passed in ready-made; only the scheduler may be read from disk."""
        if scheduler is None:
            try:
                scheduler = DDIMScheduler.from_pretrained(model_dir, subfolder="scheduler")
            except (FileNotFoundError, NotADirectoryError):
                scheduler = DDIMScheduler()
        pipe = cls(tokenizer, text_encoder, vae, transformer, scheduler)
        return pipe.to(dtype=torch_dtype) if torch_dtype is not None else pipe

    def to(self, device=None, dtype=None):
        for m in (self.vae, self.transformer):
            if m is not None:
                m.to(device=device, dtype=dtype)
        if self.text_encoder is not None and device is not None:
            self.text_encoder.to(device)
        return self

    def enable_model_cpu_offload(self, *a, **k):
        """demo.py:668-671 shuttles weights host<->device on every call to fit 28 GB cards; with 288 GB of
        HBM the 12 GB of weights simply stay resident -> move once, no hooks."""
        return self.to("cuda")

    enable_sequential_cpu_offload = enable_model_cpu_offload

    def maybe_free_model_hooks(self):
        pass

    @property
    def device(self):
        return self.transformer.device

    @property
    def dtype(self):
        return self.transformer.dtype

    _execution_device = device

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def interrupt(self):
        return self._interrupt

    # ---- prompt encoding (:248-381): conditioning I/O, runs the caller's T5 if one was given ----
    def _get_t5_prompt_embeds(self, prompt=None, num_videos_per_prompt: int = 1, max_sequence_length: int = 226, device=None,
                              dtype=None):
        """reference :248-296: tokenizer (padded / truncated to `max_sequence_length`) -> text encoder -> [B,L,D]."""
        if self.text_encoder is None or self.tokenizer is None:
            raise ValueError("no text encoder: pass `prompt_embeds` / `negative_prompt_embeds` instead of `prompt`")
        device = device or self._execution_device
        dtype = dtype or self.text_encoder.dtype
        prompt = [prompt] if isinstance(prompt, str) else prompt
        batch_size = len(prompt)
        text_input_ids = self.tokenizer(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                                        add_special_tokens=True, return_tensors="pt").input_ids
        untruncated_ids = self.tokenizer(prompt, padding="longest", return_tensors="pt").input_ids
        if untruncated_ids.shape[-1] >= text_input_ids.shape[-1] and not torch.equal(text_input_ids, untruncated_ids):
            removed_text = self.tokenizer.batch_decode(untruncated_ids[:, max_sequence_length - 1:-1])
            print("The following part of your input was truncated because `max_sequence_length` is set to "
                  f" {max_sequence_length} tokens: {removed_text}")
        prompt_embeds = self.text_encoder(text_input_ids.to(device))[0].to(dtype=dtype, device=device)
        _, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_videos_per_prompt, 1)
        return prompt_embeds.view(batch_size * num_videos_per_prompt, seq_len, -1)

    def encode_prompt(self, prompt, negative_prompt=None, do_classifier_free_guidance=True, num_videos_per_prompt=1,
                      prompt_embeds=None, negative_prompt_embeds=None, max_sequence_length=226, device=None, dtype=None):
        """reference :298-381."""
        device = device or self._execution_device
        prompt = [prompt] if isinstance(prompt, str) else prompt
        batch_size = len(prompt) if prompt is not None else prompt_embeds.shape[0]
        if prompt_embeds is None:
            prompt_embeds = self._get_t5_prompt_embeds(prompt, num_videos_per_prompt, max_sequence_length, device, dtype)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            negative_prompt = negative_prompt or ""
            negative_prompt = batch_size * [negative_prompt] if isinstance(negative_prompt, str) else negative_prompt
            if prompt is not None and type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                                 f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                                 " the batch size of `prompt`.")
            negative_prompt_embeds = self._get_t5_prompt_embeds(negative_prompt, num_videos_per_prompt, max_sequence_length,
                                                                device, dtype)
        return prompt_embeds, negative_prompt_embeds

    def check_inputs(self, prompt, height, width, negative_prompt, callback_on_step_end_tensor_inputs,
                     prompt_embeds=None, negative_prompt_embeds=None):
        """reference :543-599 (same conditions, same exception type)."""
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_on_step_end_tensor_inputs is not None and not all(
                k in self._callback_tensor_inputs for k in callback_on_step_end_tensor_inputs):
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one of the two.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        elif prompt is not None and (not isinstance(prompt, str) and not isinstance(prompt, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `negative_prompt_embeds`.")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None:
            if prompt_embeds.shape != negative_prompt_embeds.shape:
                raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, "
                                 f"but got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` "
                                 f"{negative_prompt_embeds.shape}.")

    def prepare_latents(self, batch_size, num_channels_latents, height, width, video_length, dtype, device, generator,
                        latents=None, video=None, timestep=None, is_strength_max=True, return_noise=False,
                        return_video_latents=False):
        """reference :383-457, same arguments, same return: the tuple `(latents,) [+ (noise,)] [+ (video_latents,)]`.
        strength == 1: pure noise * init_noise_sigma; strength < 1 (and no `latents=`): the VAE-encoded preprocessed `video` —
        one posterior SAMPLE per batch item, drawn item by item from the global RNG like the reference's loop (:414-421), times
        scaling_factor — noised to the first timestep of the shortened loop (`scheduler.add_noise`, :431-436)."""
        shape = (batch_size, (video_length - 1) // self.vae_scale_factor_temporal + 1, num_channels_latents,
                 height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                             f"batch size of {batch_size}. Make sure the batch size matches the length of the generators.")
        video_latents = None
        if return_video_latents or (latents is None and not is_strength_max):
            if video is None:
                raise ValueError("`strength` < 1 (or return_video_latents=True) starts from the encoded `video`: pass `video=`")
            video = video.to(device=device, dtype=dtype)
            parts = [self.vae.encode(video[i:i + 1])[0].sample() for i in range(video.shape[0])]          # :414-421, bs = 1
            vl = torch.cat(parts, dim=0) * self.vae.config.scaling_factor
            video_latents = vl.repeat(batch_size // vl.shape[0], 1, 1, 1, 1).to(device=device, dtype=dtype).permute(0, 2, 1, 3, 4)
        if latents is None:
            gdev = generator.device if generator is not None and not isinstance(generator, list) else device
            noise = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)     # randn_tensor
            # pure noise is scaled by the scheduler's init sigma; image + noise is not (:438-442)
            latents = noise * self.scheduler.init_noise_sigma if is_strength_max else \
                self.scheduler.add_noise(video_latents, noise, timestep).to(dtype)
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"`latents` has shape {tuple(latents.shape)}, expected {shape}")
            noise = latents.to(device=device, dtype=dtype)
            latents = noise * self.scheduler.init_noise_sigma
        outputs = (latents,)
        if return_noise:
            outputs += (noise,)
        if return_video_latents:
            outputs += (video_latents,)
        return outputs

    def _prepare_rotary_positional_embeddings(self, height: int, width: int, num_frames: int, device):
        """reference :616-649."""
        p = self.transformer.config.patch_size
        gh, gw = height // (self.vae_scale_factor_spatial * p), width // (self.vae_scale_factor_spatial * p)
        base_w, base_h = 720 // (self.vae_scale_factor_spatial * p), 480 // (self.vae_scale_factor_spatial * p)
        crops = get_resize_crop_region_for_grid((gh, gw), base_w, base_h)
        cos, sin = get_3d_rotary_pos_embed(self.transformer.config.attention_head_dim, crops, (gh, gw), num_frames)
        return cos.to(device), sin.to(device)

    def get_timesteps(self, num_inference_steps, strength, device=None):
        """reference :664-671."""
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start * self.scheduler.order:], num_inference_steps - t_start

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """reference :508-518 -> fp32 frames [B,3,F,H,W] in [0,1] ON THE GPU (the caller decides about .cpu())."""
        z = latents.permute(0, 2, 1, 3, 4)
        return self.vae.decode_to_frames(z, scale=1.0 / self.vae.config.scaling_factor)

    @staticmethod
    def _preprocess(x: torch.Tensor, height: int, width: int, do_normalize: bool = True, do_binarize: bool = False):
        """diffusers VaeImageProcessor.preprocess for [B,C,F,H,W] tensors (:864-870,876-879,952-960)."""
        b, c, f = x.shape[:3]
        y = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, *x.shape[3:]).float()
        if y.shape[-2:] != (height, width):
            y = F.interpolate(y, size=(height, width))
        if do_normalize and float(y.min()) >= 0:
            y = 2.0 * y - 1.0
        if do_binarize:
            y = (y >= 0.5).to(y.dtype)
        return y.reshape(b, f, c, height, width).permute(0, 2, 1, 3, 4)

    def prepare_extra_step_kwargs(self, generator, eta):
        """reference :521-540: what the loop hands `scheduler.step` besides (model_output, t, sample) — `eta` only to schedulers whose
        step takes it (the DDIM pair), `generator` to those that draw (here: every `fused_cfg_step` accepts it)."""
        kw = {"generator": generator}
        if getattr(self.scheduler, "accepts_eta", False):
            kw["eta"] = eta
        return kw

    def prepare_mask_latents(self, mask, masked_image, batch_size, height, width, dtype, device, generator,
                             do_classifier_free_guidance, noise_aug_strength):
        """reference :459-506: VAE-encode (posterior mode, times scaling_factor) a mask video and / or a masked video, one batch item
        at a time like the reference; `add_noise_in_inpaint_model` noises the masked video first.  (`__call__` does the same inside
        `_build_conditioning`, where the mask is resized instead of encoded, :991-996.)"""
        sf = self.vae.config.scaling_factor

        def enc(x):
            x = x.to(device=device, dtype=BF16)
            return torch.cat([self.vae.encode(x[i:i + 1])[0].mode() for i in range(x.shape[0])], dim=0) * sf

        mask = enc(mask) if mask is not None else None
        masked_image_latents = None
        if masked_image is not None:
            if self.transformer.config.add_noise_in_inpaint_model:
                masked_image = add_noise_to_reference_video(masked_image, ratio=noise_aug_strength)
            masked_image_latents = enc(masked_image)
        return mask, masked_image_latents

    def fuse_qkv_projections(self) -> None:
        """reference :1218-1222 (delegates to the transformer; the HIP model always runs one fused QKV GEMM)."""
        self.transformer.fuse_qkv_projections()

    def unfuse_qkv_projections(self) -> None:
        self.transformer.unfuse_qkv_projections()

    def _build_conditioning(self, video, mask_video, reference, height, width, do_cfg, dtype, device,
                            noise_aug_strength: Optional[float] = 0.0563, masked_video_latents: Optional[torch.Tensor] = None):
        """reference :862-897 and :927-1028: pixels -> (inpaint_latents [B,T,17,h,w], ref_latents [B,Tr,16,h,w]) through
        the HIP VAE encoder.  The elementwise preparation (normalise, binarise, trilinear mask resize) is
        conditioning I/O on small tensors and stays in torch."""
        if video is None or reference is None:
            raise ValueError("`video` and `reference` are required to build the conditioning (or pass `inpaint_latents=` / "
                             "`ref_latents=` directly)")
        sf = self.vae.config.scaling_factor
        init_video = self._preprocess(video.to(device), height, width)
        ref_video = self._preprocess(reference.to(device), height, width)
        ref_lat = self.vae.encode(ref_video.to(dtype))[0].sample() * sf                              # :885-889 (global RNG)
        ref_lat = ref_lat.to(dtype).permute(0, 2, 1, 3, 4)
        B, _, Fv = video.shape[:3]
        T = (Fv - 1) // self.vae_scale_factor_temporal + 1
        h, w = height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial
        if mask_video is None or bool((mask_video == 255).all()):                                      # :928-948
            mask_lat = torch.zeros(B, T, 1, h, w, device=device, dtype=dtype)
            mv_lat = torch.zeros(B, T, self.vae.config.latent_channels, h, w, device=device, dtype=dtype)
        else:
            mask_cond = self._preprocess(mask_video.to(device), height, width, do_normalize=False, do_binarize=True)
            tile = mask_cond.repeat(1, 3, 1, 1, 1)
            if masked_video_latents is None:
                masked_video = init_video * (tile < 0.5) + torch.ones_like(init_video) * (tile > 0.5) * -1   # :969-974
            else:
                # despite its name the reference treats this kwarg as a PIXEL-space masked video [B,3,F,H,W] in [-1,1] that
                # replaces the one built from `video` and `mask_video`, and encodes it like that one (:975-989)
                masked_video = masked_video_latents.to(device=device, dtype=torch.float32)
                if masked_video.shape != init_video.shape:
                    raise ValueError(f"`masked_video_latents` must be a pixel-space video {tuple(init_video.shape)} (the reference "
                                     f"encodes it with the VAE, :975-989), got {tuple(masked_video.shape)}")
            if self.transformer.config.add_noise_in_inpaint_model:                                     # :488-491
                masked_video = add_noise_to_reference_video(masked_video, ratio=noise_aug_strength)
            mv = (self.vae.encode(masked_video.to(dtype))[0].mode() * sf).to(dtype)                    # :498-502
            mask_lat = (resize_mask(1 - mask_cond, mv) * sf).to(dtype).permute(0, 2, 1, 3, 4)          # :991-996
            mv_lat = mv.permute(0, 2, 1, 3, 4)
        inpaint = torch.cat([mask_lat, mv_lat], dim=2).contiguous()                                    # :1026-1028
        return inpaint, ref_lat.contiguous()

    @torch.no_grad()
    def __call__(
        self,
        prompt: Optional[Union[str, List[str]]] = None,
        negative_prompt: Optional[Union[str, List[str]]] = None,
        height: int = 480,
        width: int = 720,
        video: Optional[torch.Tensor] = None,
        mask_video: Optional[torch.Tensor] = None,
        reference: Optional[torch.Tensor] = None,
        masked_video_latents: Optional[torch.Tensor] = None,
        num_frames: int = 49,
        num_inference_steps: int = 50,
        timesteps: Optional[List[int]] = None,
        guidance_scale: float = 6,
        use_dynamic_cfg: bool = False,
        num_videos_per_prompt: int = 1,
        eta: float = 0.0,
        generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
        latents: Optional[torch.Tensor] = None,
        prompt_embeds: Optional[torch.Tensor] = None,
        negative_prompt_embeds: Optional[torch.Tensor] = None,
        output_type: str = "numpy",
        return_dict: bool = False,
        callback_on_step_end: Optional[Callable] = None,
        callback_on_step_end_tensor_inputs: List[str] = ["latents"],
        max_sequence_length: int = 226,
        strength: float = 1,
        noise_aug_strength: float = 0.0563,
        comfyui_progressbar: bool = False,
        inpaint_latents: Optional[torch.Tensor] = None,
        ref_latents: Optional[torch.Tensor] = None,
    ) -> CogVideoX_Fun_PipelineOutput:
        st = self.prepare_denoise(
            prompt=prompt, negative_prompt=negative_prompt, height=height, width=width, video=video, mask_video=mask_video,
            reference=reference, num_frames=num_frames, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
            use_dynamic_cfg=use_dynamic_cfg, eta=eta, generator=generator, latents=latents, prompt_embeds=prompt_embeds,
            negative_prompt_embeds=negative_prompt_embeds, callback_on_step_end_tensor_inputs=callback_on_step_end_tensor_inputs,
            max_sequence_length=max_sequence_length, strength=strength, noise_aug_strength=noise_aug_strength,
            inpaint_latents=inpaint_latents, ref_latents=ref_latents, masked_video_latents=masked_video_latents)
        # `timesteps=`: accepted and never read, exactly like the reference (declared :686, but :846 calls
        # `scheduler.set_timesteps(num_inference_steps)` and `retrieve_timesteps` is never used).
        pbar = None
        if comfyui_progressbar:                                                      # :851-854 (ComfyUI node integration)
            from comfy.utils import ProgressBar
            pbar = ProgressBar(st.num_inference_steps + 2)
            pbar.update(2)                                                           # :862, :925: conditioning + latents ready

        # 8. denoising loop (:1089-1198)
        ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        ev0.record()
        for i, t in enumerate(st.timesteps):
            if self.interrupt:
                continue
            self.denoise_step(st, t)
            if callback_on_step_end is not None:                                       # :1181-1190
                latents, prompt_embeds = st.latents, st.prompt_embeds
                negative_prompt_embeds = st.prompt_embeds[:st.batch_size] if st.do_cfg else None
                callback_kwargs = {k: locals()[k] for k in callback_on_step_end_tensor_inputs}
                callback_outputs = callback_on_step_end(self, i, t, callback_kwargs)
                st.latents = callback_outputs.pop("latents", st.latents)
                st.prompt_embeds = callback_outputs.pop("prompt_embeds", st.prompt_embeds)
            if pbar is not None:                                                     # :1197-1198
                pbar.update(1)
        ev1.record()

        if output_type == "latent":
            video_out = st.latents
        elif output_type == "cl_bf16":
            # the decoder's own bf16 output, channels-last [B,F,H,W,3], before the frame conversion of decode_latents: what the
            # data-parallel runner all-gathers (driver.run_orbits); `vae.cl_to_frames` of it == decode_latents, bit for bit
            video_out = self.vae.decode_cl_bf16(st.latents.permute(0, 2, 1, 3, 4), scale=1.0 / self.vae.config.scaling_factor)
        else:
            video_out = self.decode_latents(st.latents)
        ev2.record()
        if output_type in ("numpy", "np"):
            video_out = video_out.cpu()                 # reference :517 / :1214 returns a CPU float tensor
        self._events = (ev0, ev1, ev2)
        self.maybe_free_model_hooks()
        return CogVideoX_Fun_PipelineOutput(videos=video_out)

    # ---- the loop body and its set-up as separate entry points: `__call__` above is prepare -> N x step -> decode; bench.py
    #      times `denoise_step` itself, so the measured step IS the product step ----
    @torch.no_grad()
    def prepare_denoise(self, prompt=None, negative_prompt=None, height=480, width=720, video=None, mask_video=None,
                        reference=None, num_frames=49, num_inference_steps=50, guidance_scale=6, use_dynamic_cfg=False,
                        eta=0.0, generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                        callback_on_step_end_tensor_inputs=("latents",), max_sequence_length=226, strength=1,
                        noise_aug_strength=0.0563, inpaint_latents=None, ref_latents=None,
                        masked_video_latents=None) -> "DenoiseState":
        """Everything of reference `__call__` before the loop (:786-1087): checks, prompt embeddings, timesteps,
        conditioning latents, initial noise, rotary tables."""
        if num_frames > 49:
            raise ValueError("The number of frames must be less than 49 for now due to static positional embeddings. "
                             "This will be updated in the future to remove this limitation.")
        if not 0.0 <= eta <= 1.0:
            raise ValueError(f"`eta` must be in [0, 1], got {eta}")
        if not 0.0 < strength <= 1.0:
            raise ValueError(f"`strength` must be in (0, 1], got {strength}")
        if strength < 1.0 and type(self.scheduler).__name__ == "PNDMScheduler":
            raise NotImplementedError("`strength` < 1 with the PNDM sampler is not built: its schedule is stateful (12 Runge-Kutta evaluations "
                                      "first) and the shortened loop of :664-671 would enter it mid-way; use DDIM_Origin / DDIM_Cog / Euler / "
                                      "Euler A / DPM++ for strength < 1")
        num_videos_per_prompt = 1
        if hasattr(self.transformer, "clear_cross_kv_cache"):
            self.transformer.clear_cross_kv_cache()               # opt-in K / V reuse (model.cache_cross_kv) never outlives a clip
        self.check_inputs(prompt, height, width, negative_prompt, list(callback_on_step_end_tensor_inputs), prompt_embeds,
                          negative_prompt_embeds)
        self._guidance_scale = guidance_scale
        self._interrupt = False
        device = self.device
        if device.type != "cuda" or self.dtype != BF16:
            raise TcxError(f"TrajCrafter_Pipeline: models must be bf16 on the GPU (got {self.dtype} on {device}); "
                           "the HIP path has no CPU fallback")
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = guidance_scale > 1.0

        # 3. prompt (:831-843)
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, negative_prompt, do_cfg, num_videos_per_prompt, prompt_embeds, negative_prompt_embeds,
            max_sequence_length, device)
        prompt_embeds = prompt_embeds.to(device=device, dtype=BF16)
        if do_cfg:
            prompt_embeds = torch.cat([negative_prompt_embeds.to(device=device, dtype=BF16), prompt_embeds], dim=0)

        # 4. timesteps (:846-850): host ints
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps, num_inference_steps = self.get_timesteps(num_inference_steps, strength, device)
        self._num_timesteps = len(timesteps)

        # 5. conditioning + latents (:862-1068)
        if inpaint_latents is None or ref_latents is None:
            inpaint_latents, ref_latents = self._build_conditioning(video, mask_video, reference, height, width, do_cfg,
                                                                    BF16, device, noise_aug_strength, masked_video_latents)
        elif masked_video_latents is not None:
            raise ValueError("`masked_video_latents` only takes part in building the conditioning from pixels; it cannot be "
                             "combined with pre-encoded `inpaint_latents=` / `ref_latents=`")
        video_length = video.shape[2] if video is not None else num_frames           # quirk: real count = video.shape[2]
        rep = 2 if do_cfg else 1
        inpaint_latents = inpaint_latents.to(device=device, dtype=BF16)
        ref_input = ref_latents.to(device=device, dtype=BF16)
        if inpaint_latents.shape[0] == batch_size and rep == 2:
            inpaint_latents = torch.cat([inpaint_latents] * 2)
        if ref_input.shape[0] == batch_size and rep == 2:
            ref_input = torch.cat([ref_input] * 2)
        num_channels_latents = self.vae.config.latent_channels
        init_video = None
        if strength != 1 and latents is None and video is not None:
            init_video = self._preprocess(video.to(device), height, width)                              # :863-871
        latents = self.prepare_latents(batch_size * num_videos_per_prompt, num_channels_latents, height, width,
                                       video_length, BF16, device, generator, latents, video=init_video,
                                       timestep=timesteps[:1], is_strength_max=strength == 1)[0]
        latents = latents.contiguous()
        if inpaint_latents.shape[:2] != (rep * batch_size, latents.shape[1]) or inpaint_latents.shape[3:] != latents.shape[3:]:
            raise ValueError(f"inpaint_latents {tuple(inpaint_latents.shape)} does not match latents {tuple(latents.shape)}")

        # 7. rotary tables (:1076-1082)
        image_rotary_emb = (self._prepare_rotary_positional_embeddings(height, width, latents.size(1), device)
                            if self.transformer.config.use_rotary_positional_embeddings else None)
        return DenoiseState(latents=latents, prompt_embeds=prompt_embeds, inpaint_latents=inpaint_latents.contiguous(),
                            ref_input=ref_input.contiguous(), image_rotary_emb=image_rotary_emb,
                            timesteps=[int(t) for t in timesteps.tolist()], num_inference_steps=num_inference_steps,
                            batch_size=batch_size, do_cfg=do_cfg, guidance_scale=float(guidance_scale),
                            use_dynamic_cfg=use_dynamic_cfg, generator=generator if not isinstance(generator, list) else generator[0],
                            eta=float(eta))

    @torch.no_grad()
    def denoise_step(self, st: "DenoiseState", t: int) -> torch.Tensor:
        """One iteration of the reference loop (:1093-1178): CFG-batched transformer forward, guidance, DDIM update,
        bf16 latents.  Updates and returns `st.latents`.  No host synchronisation."""
        device = st.latents.device
        lat_in = self.scheduler.scale_model_input(st.latents, t)                   # :1099-1101 (identity for the DDIM samplers)
        latent_model_input = torch.cat([lat_in] * 2) if st.do_cfg else lat_in
        timestep = torch.full((latent_model_input.shape[0],), t, device=device, dtype=torch.int64)
        noise_pred = self.transformer(hidden_states=latent_model_input, encoder_hidden_states=st.prompt_embeds,
                                      timestep=timestep, image_rotary_emb=st.image_rotary_emb, return_dict=False,
                                      inpaint_latents=st.inpaint_latents, cross_latents=st.ref_input)[0]
        if st.use_dynamic_cfg:                                                     # :1142-1156
            n = st.num_inference_steps
            self._guidance_scale = 1 + st.guidance_scale * ((1 - math.cos(math.pi * ((n - t) / n) ** 5.0)) / 2)
        skw = {"generator": st.generator}                                          # prepare_extra_step_kwargs (:521-540)
        if st.eta and getattr(self.scheduler, "accepts_eta", False):
            skw["eta"] = st.eta
        if st.do_cfg:                                                              # :1157-1178 fused, per scheduler class
            u, c = noise_pred[:st.batch_size], noise_pred[st.batch_size:]
            st.latents = self.scheduler.fused_cfg_step(u, c, st.latents, self.guidance_scale, t, **skw)
        else:
            st.latents = self.scheduler.fused_cfg_step(noise_pred, None, st.latents, 1.0, t, **skw)
        return st.latents

    def timings(self) -> Dict[str, float]:
        """Seconds spent in the denoise loop and in the VAE decode of the last call (syncs)."""
        ev0, ev1, ev2 = self._events
        ev2.synchronize()
        return {"denoise_s": ev0.elapsed_time(ev1) / 1e3, "decode_s": ev1.elapsed_time(ev2) / 1e3}
