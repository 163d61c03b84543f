"""Restatement of the `diffusers` arithmetic the reference's hot path calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  `diffusers` (requirements.txt:26,
``diffusers>=0.30.1``, unpinned) is a third-party dependency that is NOT vendored
in /root/reference and NOT installed in this image, so every function here is a
restatement of its *published* algorithm -> **parity unpinned**; the call sites in
the reference that anchor each function are cited.  Known-answer tests:
tests/test_oracle_kat.py.

All functions take / return float32 working tensors; `p` is an oracle.prec.Prec.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .prec import Prec

LOG2E = 1.4426950408889634


# ---------------------------------------------------------------------------
# embeddings  (call sites: reference models/crosstransformer3d.py:531-534,724-732)
# ---------------------------------------------------------------------------
def timesteps_proj(timesteps: torch.Tensor, num_channels: int, flip_sin_to_cos: bool = True,
                   downscale_freq_shift: float = 0.0, scale: float = 1.0,
                   max_period: int = 10000) -> torch.Tensor:
    """diffusers `Timesteps` / get_timestep_embedding: fp32 sinusoidal embedding."""
    assert timesteps.ndim == 1
    half = num_channels // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if num_channels % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def timestep_embedding(p: Prec, sd: dict, prefix: str, t_emb: torch.Tensor) -> torch.Tensor:
    """diffusers `TimestepEmbedding(in, time_embed_dim, "silu")`: linear_2(silu(linear_1(x)))."""
    x = p.linear(t_emb, sd[prefix + "linear_1.weight"], sd[prefix + "linear_1.bias"])
    x = p.r(F.silu(x))
    return p.linear(x, sd[prefix + "linear_2.weight"], sd[prefix + "linear_2.bias"])


# ---------------------------------------------------------------------------
# normalisation  (call sites: crosstransformer3d.py:195-197,211-213,556-562)
# ---------------------------------------------------------------------------
def layer_norm_zero(p: Prec, sd: dict, prefix: str, hidden, encoder, temb, eps: float):
    """diffusers `CogVideoXLayerNormZero.forward`.

    s = linear(silu(temb)); (shift, scale, gate, e_shift, e_scale, e_gate) = s.chunk(6, 1)
    video rows: LN(x)*(1+scale)+shift ; text rows: LN(x)*(1+e_scale)+e_shift ; returns gates [B,1,D].
    """
    s = p.linear(p.r(F.silu(temb)), sd[prefix + "linear.weight"], sd[prefix + "linear.bias"])
    shift, scale, gate, e_shift, e_scale, e_gate = s.chunk(6, dim=1)
    w, b = sd.get(prefix + "norm.weight"), sd.get(prefix + "norm.bias")

    def mod(x, sc, sh):
        n = p.layer_norm(x, w, b, eps)                       # per-op point in the reference
        y = p.r(n * p.r(1 + sc)[:, None, :])
        return p.R(y + sh[:, None, :])                       # contract point: fused LN+modulate output

    return mod(hidden, scale, shift), mod(encoder, e_scale, e_shift), gate[:, None, :], e_gate[:, None, :]


def ada_layer_norm(p: Prec, sd: dict, prefix: str, x, temb, eps: float):
    """diffusers `AdaLayerNorm(chunk_dim=1)`: **shift first**: shift, scale = t.chunk(2, 1)."""
    t = p.linear(p.r(F.silu(temb)), sd[prefix + "linear.weight"], sd[prefix + "linear.bias"])
    shift, scale = t.chunk(2, dim=1)
    n = p.layer_norm(x, sd.get(prefix + "norm.weight"), sd.get(prefix + "norm.bias"), eps)
    y = p.r(n * p.r(1 + scale)[:, None, :])
    return p.R(y + shift[:, None, :])


# ---------------------------------------------------------------------------
# rotary embedding (call sites: pipeline_trajectorycrafter.py:616-649; processor below)
# ---------------------------------------------------------------------------
def get_3d_rotary_pos_embed(embed_dim: int, crops_coords, grid_size, temporal_size: int,
                            theta: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """diffusers `get_3d_rotary_pos_embed(..., use_real=True)` -> (cos, sin) each [T*H*W, embed_dim] fp32."""
    start, stop = crops_coords
    gh, gw = grid_size
    grid_h = np.linspace(start[0], stop[0], gh, endpoint=False, dtype=np.float32)
    grid_w = np.linspace(start[1], stop[1], gw, endpoint=False, dtype=np.float32)
    grid_t = np.linspace(0, temporal_size, temporal_size, endpoint=False, dtype=np.float32)

    dim_t = embed_dim // 4
    dim_h = embed_dim // 8 * 3
    dim_w = embed_dim // 8 * 3

    def axis(pos, dim):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        ang = torch.outer(torch.from_numpy(pos).float(), freqs)
        return ang.repeat_interleave(2, dim=-1)

    ft, fh, fw = axis(grid_t, dim_t), axis(grid_h, dim_h), axis(grid_w, dim_w)
    T = temporal_size
    ft = ft[:, None, None, :].expand(T, gh, gw, dim_t)
    fh = fh[None, :, None, :].expand(T, gh, gw, dim_h)
    fw = fw[None, None, :, :].expand(T, gh, gw, dim_w)
    freqs = torch.cat([ft, fh, fw], dim=-1).reshape(T * gh * gw, -1)
    return freqs.cos().contiguous(), freqs.sin().contiguous()


def get_1d_sincos_pos_embed_from_grid(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """diffusers `get_1d_sincos_pos_embed_from_grid`: [sin(pos w_i) | cos(pos w_i)], w_i = 10000^(-i / (embed_dim/2)), float64."""
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_3d_sincos_pos_embed(embed_dim: int, spatial_size, temporal_size: int, spatial_interpolation_scale: float = 1.0,
                            temporal_interpolation_scale: float = 1.0) -> np.ndarray:
    """diffusers `get_3d_sincos_pos_embed` (numpy version, the one the reference's `torch.from_numpy` at crosstransformer3d.py:516-523
    implies) -> [T, H*W, embed_dim]: a quarter of the channels encode the frame, three quarters the (h | w) position.  The caller
    passes spatial_size = (post_patch_width, post_patch_height) (:518).  PARITY UNPINNED: diffusers is absent offline."""
    if embed_dim % 4 != 0:
        raise ValueError("`embed_dim` must be divisible by 4")
    if isinstance(spatial_size, int):
        spatial_size = (spatial_size, spatial_size)
    dim_spatial, dim_temporal = 3 * embed_dim // 4, embed_dim // 4
    grid_h = np.arange(spatial_size[1], dtype=np.float32) / spatial_interpolation_scale
    grid_w = np.arange(spatial_size[0], dtype=np.float32) / spatial_interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0)                 # w first
    grid = grid.reshape([2, 1, spatial_size[1], spatial_size[0]])
    emb_h = get_1d_sincos_pos_embed_from_grid(dim_spatial // 2, grid[0])
    emb_w = get_1d_sincos_pos_embed_from_grid(dim_spatial // 2, grid[1])
    pos_spatial = np.concatenate([emb_h, emb_w], axis=1)                 # [H*W, 3D/4]
    grid_t = np.arange(temporal_size, dtype=np.float32) / temporal_interpolation_scale
    pos_temporal = get_1d_sincos_pos_embed_from_grid(dim_temporal, grid_t)
    pos_spatial = np.repeat(pos_spatial[np.newaxis], temporal_size, axis=0)
    pos_temporal = np.repeat(pos_temporal[:, np.newaxis], spatial_size[0] * spatial_size[1], axis=1)
    return np.concatenate([pos_temporal, pos_spatial], axis=-1)


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """diffusers `apply_rotary_emb(x, (cos, sin), use_real=True, use_real_unbind_dim=-1)`; x [B,H,S,D] fp32."""
    cos = cos[None, None]
    sin = sin[None, None]
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return x.float() * cos + rot.float() * sin


# ---------------------------------------------------------------------------
# attention (call site: crosstransformer3d.py:199-208, 239-243)
# ---------------------------------------------------------------------------
def sdpa(p: Prec, q, k, v, scale: float) -> torch.Tensor:
    """softmax(q k^T * scale) v with fp32 scores.

    Rounding contract of the HIP flash kernel (bf16 mode): the UNNORMALISED probabilities
    e = exp(s - rowmax) are rounded to the activation dtype before PV and the row sum l is taken from
    the unrounded e in fp32 (flash-attention practice; SURVEY §8c "P rounded to bf16 before PV").
    The reference's eager execution (bf16_ref mode) rounds the normalised softmax instead
    (crosstransformer3d.py:394-395 under autocast; SDPA-flash inside diffusers).  fp32 mode: no rounding.
    """
    if q.shape[0] * q.shape[1] > 1 and q.shape[0] * q.shape[1] * q.shape[-2] * k.shape[-2] > (1 << 29):
        # bound the score matrix (full-size CPU baseline: 48 x 17776^2 fp32 would be 60 GB): one head at a time
        return torch.stack([torch.stack([sdpa(p, q[b:b + 1, h:h + 1], k[b:b + 1, h:h + 1], v[b:b + 1, h:h + 1], scale)[0, 0]
                                         for h in range(q.shape[1])]) for b in range(q.shape[0])])
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    if p.mode == "bf16_ref":
        return torch.matmul(p.R(torch.softmax(s, dim=-1)), v.float())
    e = torch.exp(s - s.amax(dim=-1, keepdim=True))
    l = e.sum(dim=-1, keepdim=True)
    return torch.matmul(p.R(e), v.float()) / l


def sdpa_log2(p: Prec, q, k, v) -> torch.Tensor:
    """Contract of the FAST self-attention HIP path (tcx_attn_fwd with TCX_ATTN_LOG2_SCORES): q arrives
    pre-multiplied by scale*log2(e), P = exp2(q k^T - rowmax) is rounded to the activation dtype and BOTH
    PV and the row sum use the rounded P (the sum runs on the matrix pipe next to PV)."""
    if q.shape[0] * q.shape[1] > 1 and q.shape[0] * q.shape[1] * q.shape[-2] * k.shape[-2] > (1 << 29):
        return torch.stack([torch.stack([sdpa_log2(p, q[b:b + 1, h:h + 1], k[b:b + 1, h:h + 1], v[b:b + 1, h:h + 1])[0, 0]
                                         for h in range(q.shape[1])]) for b in range(q.shape[0])])
    s = torch.matmul(q.float(), k.float().transpose(-1, -2))
    e = p.R(torch.exp2(s - s.amax(dim=-1, keepdim=True)))
    return torch.matmul(e, v.float()) / e.sum(dim=-1, keepdim=True)


def cogvideox_attention(p: Prec, sd: dict, prefix: str, hidden, encoder, heads: int,
                        rotary: Optional[Tuple[torch.Tensor, torch.Tensor]], eps: float = 1e-6):
    """diffusers `Attention` + `CogVideoXAttnProcessor2_0.__call__`; returns (video, text)."""
    text_len = encoder.shape[1]
    h = torch.cat([encoder, hidden], dim=1)
    B, S, D = h.shape
    dh = D // heads

    def proj(name):
        y = p.linear(h, sd[prefix + name + ".weight"], sd.get(prefix + name + ".bias"))
        return y.view(B, S, heads, dh).transpose(1, 2)

    q, k, v = proj("to_q"), proj("to_k"), proj("to_v")
    q = p.layer_norm(q, sd[prefix + "norm_q.weight"], sd[prefix + "norm_q.bias"], eps)
    k = p.layer_norm(k, sd[prefix + "norm_k.weight"], sd[prefix + "norm_k.bias"], eps)
    if rotary is not None:
        cos, sin = rotary
        q = torch.cat([q[:, :, :text_len], apply_rotary_emb(q[:, :, text_len:], cos, sin)], dim=2)
        k = torch.cat([k[:, :, :text_len], apply_rotary_emb(k[:, :, text_len:], cos, sin)], dim=2)
    if p.mode == "bf16":
        # contract of the HIP path: q is stored pre-multiplied by dh^-1/2 * log2(e) (one rounding, in the fused
        # qk-LN+RoPE kernel) and the attention kernel works on base-2 scores
        q, k = p.R(q * (dh ** -0.5 * LOG2E)), p.R(k)
        o = sdpa_log2(p, q, k, v)
    else:
        q, k = p.R(q), p.R(k)                                # reference: unscaled q, softmax(q k^T / sqrt(dh))
        o = sdpa(p, q, k, v, scale=dh ** -0.5)
    o = p.R(o).transpose(1, 2).reshape(B, S, D)              # contract: attention output
    o = p.linear_fused(o, sd[prefix + "to_out.0.weight"], sd.get(prefix + "to_out.0.bias"))   # consumed by the gated residual
    return o[:, text_len:], o[:, :text_len]


def feed_forward(p: Prec, sd: dict, prefix: str, x):
    """diffusers `FeedForward(activation_fn="gelu-approximate")`: net.0 = GELU(tanh) proj, net.2 = Linear."""
    y = F.linear(x.float(), p.param(sd[prefix + "net.0.proj.weight"]), p.param(sd[prefix + "net.0.proj.bias"]))
    y = p.r(y)                                               # reference rounds the GEMM output first
    y = p.R(F.gelu(y, approximate="tanh"))                   # contract: fused bias+GELU output
    return p.linear_fused(y, sd[prefix + "net.2.weight"], sd[prefix + "net.2.bias"])      # consumed by the gated residual


# ---------------------------------------------------------------------------
# VAE helpers (call sites: autoencoder_magvit.py:428,623,1197,1212)
# ---------------------------------------------------------------------------
def upsample3d_nearest(x: torch.Tensor, compress_time: bool) -> torch.Tensor:
    """The interpolation half of diffusers `CogVideoXUpsample3D.forward` (x [B,C,T,H,W])."""
    if compress_time:
        T = x.shape[2]
        if T > 1 and T % 2 == 1:
            first, rest = x[:, :, 0], x[:, :, 1:]
            first = F.interpolate(first, scale_factor=2.0)
            rest = F.interpolate(rest, scale_factor=2.0)
            return torch.cat([first[:, :, None], rest], dim=2)
        if T > 1:
            return F.interpolate(x, scale_factor=2.0)
        return F.interpolate(x.squeeze(2), scale_factor=2.0)[:, :, None]
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.interpolate(y, scale_factor=2.0)
    return y.reshape(b, t, c, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)


def conv2d_per_frame(p: Prec, x, w, b, stride: int, padding: int):
    bsz, c, t, h, wd = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(bsz * t, c, h, wd)
    y = p.R(F.conv2d(y.float(), p.param(w), p.param(b), stride=stride, padding=padding))
    return y.reshape(bsz, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def upsample3d(p: Prec, sd: dict, prefix: str, x, compress_time: bool):
    """diffusers `CogVideoXUpsample3D(C, C, padding=1, compress_time)`: nearest x2 then per-frame Conv2d 3x3."""
    y = upsample3d_nearest(x, compress_time)
    return conv2d_per_frame(p, y, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], 1, 1)


def downsample3d(p: Prec, sd: dict, prefix: str, x, compress_time: bool):
    """diffusers `CogVideoXDownsample3D(C, C, padding=0, compress_time)` (encoder only)."""
    if compress_time:
        b, c, t, h, w = x.shape
        y = x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, t)
        if t % 2 == 1:
            first, rest = y[..., 0], y[..., 1:]
            if rest.shape[-1] > 0:
                rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
            y = torch.cat([first[..., None], rest], dim=-1)
        else:
            y = F.avg_pool1d(y, kernel_size=2, stride=2)
        y = p.R(y)
        x = y.reshape(b, h, w, c, y.shape[-1]).permute(0, 3, 4, 1, 2)
    x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    return conv2d_per_frame(p, x, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], 2, 0)


class DiagonalGaussian:
    """diffusers `DiagonalGaussianDistribution`."""

    def __init__(self, params: torch.Tensor):
        self.mean, self.logvar = torch.chunk(params, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


# ---------------------------------------------------------------------------
# image processors (call sites: pipeline_trajectorycrafter.py:237-243,864,876,952)
# ---------------------------------------------------------------------------
def vae_image_preprocess(x: torch.Tensor, height: int, width: int, do_normalize: bool = True,
                         do_binarize: bool = False) -> torch.Tensor:
    """diffusers `VaeImageProcessor.preprocess` for 4-D tensor input [N,C,H,W]."""
    if x.shape[-2:] != (height, width):
        x = F.interpolate(x, size=(height, width))
    x = x.clone()
    if do_normalize and x.min() >= 0:
        x = 2.0 * x - 1.0
    if do_binarize:
        x[x < 0.5] = 0
        x[x >= 0.5] = 1
    return x


# ---------------------------------------------------------------------------
# scheduler (selected demo.py:647-657; used pipeline_trajectorycrafter.py:846,1164-1167)
# ---------------------------------------------------------------------------
class DDIMScheduler:
    """diffusers `DDIMScheduler` restricted to the configuration CogVideoX-Fun ships
    (recalled, the checkpoint's scheduler_config.json is absent -> unverified)."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", prediction_type="v_prediction",
                 timestep_spacing="trailing", rescale_betas_zero_snr=True, set_alpha_to_one=True,
                 steps_offset=0, clip_sample=False):
        assert beta_schedule == "scaled_linear" and not clip_sample
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.timestep_spacing = timestep_spacing
        self.steps_offset = steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        if rescale_betas_zero_snr:
            alphas = 1.0 - betas
            abar_sqrt = torch.cumprod(alphas, dim=0).sqrt()
            a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
            abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
            abar = abar_sqrt ** 2
            alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
            betas = 1 - alphas
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        N = self.num_train_timesteps
        if self.timestep_spacing == "trailing":
            ts = np.round(np.arange(N, 0, -N / num_inference_steps)).astype(np.int64) - 1
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, num_inference_steps) * (N // num_inference_steps)).round()[::-1].copy().astype(np.int64)
            ts += self.steps_offset
        else:
            raise ValueError(self.timestep_spacing)
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, p: Prec, sample, timestep=None):
        return sample

    def coeffs(self, timestep: int):
        prev = timestep - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def add_noise(self, p: Prec, original_samples: torch.Tensor, noise: torch.Tensor, timestep: int) -> torch.Tensor:
        """`DDIMScheduler.add_noise`: alphas_cumprod is cast to the sample dtype and all arithmetic runs in it (every op of the
        eager bf16 execution rounds; in fp32 mode nothing does)."""
        dt = p.act_dtype
        abar = self.alphas_cumprod.to(dt)[int(timestep)]
        sa, sb = abar ** 0.5, (1 - abar) ** 0.5
        return (sa * original_samples.to(dt) + sb * noise.to(dt)).float()

    def step(self, p: Prec, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, eta: float = 0.0,
             variance_noise: Optional[torch.Tensor] = None):
        """`DDIMScheduler.step`.  `model_output` fp32, `sample` in the activation dtype.  eta > 0: std = eta sqrt((1 - a_prev)/(1 - a_t)
        (1 - a_t/a_prev)), the direction term shrinks to sqrt(1 - a_prev - std^2) eps and std * variance_noise (the library's fp32
        randn_tensor draw, supplied by the caller) is added.

        Type-promotion quirk reproduced: a 0-dim fp32 scalar times a bf16 tensor
        stays bf16, so sqrt(a)*sample and sqrt(1-a)*sample are rounded to bf16 before they
        meet the fp32 model output.
        """
        a_t, a_prev = self.coeffs(int(timestep))
        b_t = 1 - a_t
        sa, sb = a_t ** 0.5, b_t ** 0.5
        s = sample.float()
        if self.prediction_type == "v_prediction":
            x0 = p.R(sa * s) - sb * model_output
            eps = sa * model_output + p.R(sb * s)
        elif self.prediction_type == "epsilon":
            x0 = (s - sb * model_output) / sa
            eps = model_output
        else:
            raise ValueError(self.prediction_type)
        variance = ((1 - a_prev) / b_t) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        direction = torch.clamp(1 - a_prev - std ** 2, min=0) ** 0.5 * eps      # clamp: exactly 0 for eta = 1 at a zero-SNR step, -1 ulp in fp32
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            prev = prev + std * variance_noise
        return prev


class CogVideoXDDIMScheduler(DDIMScheduler):
    """diffusers `CogVideoXDDIMScheduler` (the reference's "DDIM_Cog", demo.py:652), restated from the published algorithm
    (parity unpinned like the rest of this file): float64 schedule, `alphas_cumprod / (s + (1 - s) alphas_cumprod)` SNR shift,
    then `rescale_zero_terminal_snr` on alphas_cumprod itself; `step` in the a_t / b_t form:

        x0 = sqrt(a_t) x - sqrt(1 - a_t) v;   a = sqrt((1 - a_prev) / (1 - a_t));   b = sqrt(a_prev) - sqrt(a_t) a
        x_prev = a x + b x0

    Defaults as scheduler.CogVideoXDDIMScheduler (CogVideoX-5b scheduler_config.json as recalled)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="v_prediction", timestep_spacing="trailing", rescale_betas_zero_snr=True,
                 set_alpha_to_one=True, steps_offset=0, clip_sample=False, snr_shift_scale=1.0):
        super().__init__(num_train_timesteps, beta_start, beta_end, beta_schedule, prediction_type, timestep_spacing,
                         rescale_betas_zero_snr, set_alpha_to_one, steps_offset, clip_sample)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        alphas_cumprod = alphas_cumprod / (snr_shift_scale + (1 - snr_shift_scale) * alphas_cumprod)
        if rescale_betas_zero_snr:
            alphas_bar_sqrt = alphas_cumprod.sqrt()
            a0, aT = alphas_bar_sqrt[0].clone(), alphas_bar_sqrt[-1].clone()
            alphas_bar_sqrt = alphas_bar_sqrt - aT
            alphas_bar_sqrt = alphas_bar_sqrt * (a0 / (a0 - aT))
            alphas_cumprod = alphas_bar_sqrt ** 2
        self.betas = betas
        self.alphas_cumprod = alphas_cumprod
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64) if set_alpha_to_one else alphas_cumprod[0]

    def step(self, p: Prec, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, eta: float = 0.0):
        """0-dim float64 scalars times a bf16 tensor stay bf16 (the same promotion quirk as DDIMScheduler.step): sqrt(a_t) x and
        a x are rounded before they meet the fp32 terms."""
        assert eta == 0.0
        a_t, a_prev = self.coeffs(int(timestep))
        s = sample.float()
        f = lambda v: float(v)                               # fp64 scalar -> the fp32 op-math scalar torch multiplies with
        if self.prediction_type == "v_prediction":
            x0 = p.R(f(a_t ** 0.5) * s) - f((1 - a_t) ** 0.5) * model_output
        elif self.prediction_type == "epsilon":
            x0 = (s - f((1 - a_t) ** 0.5) * model_output) / f(a_t ** 0.5)
        else:
            raise ValueError(self.prediction_type)
        a = ((1 - a_prev) / (1 - a_t)) ** 0.5
        b = a_prev ** 0.5 - a_t ** 0.5 * a
        return p.R(f(a) * s) + f(b) * x0


# ---------------------------------------------------------------------------
# sigma-parametrised samplers of the reference's table (demo.py:647-654: "Euler", "Euler A", "DPM++"), restated from the
# published diffusers algorithms (>= 0.30.1, requirements.txt:26).  PARITY UNPINNED: diffusers is absent offline; pinned only by
# the analytic known-answer tests in tests/test_oracle_kat.py.  `from_pretrained(model, subfolder="scheduler")` hands them the
# CogVideoX scheduler_config.json: scaled_linear betas 0.00085..0.012, v_prediction, trailing spacing, zero-terminal-SNR rescale
# (these classes then set alphas_cumprod[-1] = 2^-24 so that sigma stays finite: sigma_max = 4096); keys they do not know
# (snr_shift_scale, set_alpha_to_one, clip_sample ...) are ignored.  Class defaults otherwise: linear sigma interpolation, no
# Karras sigmas, final sigma 0, discrete timesteps, DPM-Solver++ 2M midpoint with lower_order_final.
# Every coefficient is a 0-dim fp32 torch tensor computed by the same operation sequence as the library's, so that the
# per-element arithmetic (fp32 mul / div / add, no contraction) can be reproduced bit for bit by the HIP step kernel.
# ---------------------------------------------------------------------------
class _SigmaScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="v_prediction", timestep_spacing="trailing", rescale_betas_zero_snr=True, steps_offset=0, **ignored):
        assert beta_schedule == "scaled_linear" and prediction_type == "v_prediction"
        self.num_train_timesteps, self.timestep_spacing, self.steps_offset = num_train_timesteps, timestep_spacing, steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        if rescale_betas_zero_snr:
            alphas = 1.0 - betas
            abar_sqrt = torch.cumprod(alphas, dim=0).sqrt()
            a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
            abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
            abar = abar_sqrt ** 2
            alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
            betas = 1 - alphas
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        if rescale_betas_zero_snr:
            self.alphas_cumprod[-1] = 2 ** -24               # "close to 0 without being 0 so the first sigma is not inf"
        self.sigmas = None
        self.timesteps = None
        self.num_inference_steps = None

    def _grid(self, n: int) -> np.ndarray:
        N = self.num_train_timesteps
        if self.timestep_spacing == "trailing":
            return np.round(np.arange(N, 0, -N / n)) - 1
        if self.timestep_spacing == "leading":
            return (np.arange(0, n) * (N // n)).round()[::-1].copy().astype(np.float64) + self.steps_offset
        raise ValueError(self.timestep_spacing)

    def _interp_sigmas(self, ts: np.ndarray) -> np.ndarray:
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        return np.interp(ts, np.arange(0, len(sig)), sig)

    def index_for(self, timestep) -> int:
        idx = (self.timesteps == timestep).nonzero()
        assert len(idx) == 1, "duplicate timesteps: the library's begin-index rule is not restated"
        return int(idx[0])

    def scale_model_input(self, p: Prec, sample: torch.Tensor, timestep) -> torch.Tensor:
        return sample

    def add_noise(self, p: Prec, original_samples: torch.Tensor, noise: torch.Tensor, timestep) -> torch.Tensor:
        """`add_noise` of the sigma samplers (the `strength < 1` start, pipeline :431-436): the sigma table is cast to the SAMPLE dtype
        and every operation runs in it.  Euler pair: x0 + noise sigma.  DPM++: alpha x0 + sig noise with alpha = 1 / sqrt(sigma^2 + 1),
        sig = sigma alpha evaluated in that dtype too (each op of the eager bf16 execution rounds)."""
        dt = p.act_dtype
        sigma = self.sigmas.to(dt)[self.index_for(timestep)]
        x0, nz = original_samples.to(dt), noise.to(dt)
        if isinstance(self, DPMSolverMultistepScheduler):
            alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
            sigma_t = sigma * alpha_t
            return (alpha_t * x0 + sigma_t * nz).float()
        return (x0 + nz * sigma).float()


class EulerDiscreteScheduler(_SigmaScheduler):
    """diffusers `EulerDiscreteScheduler` ("Euler"), s_churn = 0:

        sigma_i from the interpolated table (+ a final 0);  model input = x / sqrt(sigma^2 + 1);  init_noise_sigma = sigma_max
        x0 = v * (-sigma / sqrt(sigma^2 + 1)) + x / (sigma^2 + 1);   d = (x - x0) / sigma;   x_next = x + d (sigma_next - sigma)

    with x upcast to fp32 first and the result cast to the model output's dtype (fp32 in the reference's loop, :1108-1167)."""
    ancestral = False

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = self._grid(num_inference_steps).astype(np.float32)
        sig = np.concatenate([self._interp_sigmas(ts), [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(ts)

    @property
    def init_noise_sigma(self):
        mx = self.sigmas.max() if self.sigmas is not None else (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).max()
        return mx if self.timestep_spacing in ("linspace", "trailing") else (mx ** 2 + 1) ** 0.5

    def scale_model_input(self, p: Prec, sample: torch.Tensor, timestep) -> torch.Tensor:
        sigma = self.sigmas[self.index_for(timestep)]
        return p.R(sample / ((sigma ** 2 + 1) ** 0.5))       # bf16 tensor / 0-dim fp32 tensor stays bf16

    def step_coeffs(self, timestep):
        i = self.index_for(timestep)
        sigma, sigma_to = self.sigmas[i], self.sigmas[i + 1]
        a = -sigma / (sigma ** 2 + 1) ** 0.5
        s2p1 = sigma ** 2 + 1
        if not self.ancestral:
            return a, s2p1, sigma, sigma_to - sigma, None
        sigma_up = (sigma_to ** 2 * (sigma ** 2 - sigma_to ** 2) / sigma ** 2) ** 0.5
        sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
        return a, s2p1, sigma, sigma_down - sigma, sigma_up

    def step(self, p: Prec, model_output: torch.Tensor, timestep, sample: torch.Tensor, noise: Optional[torch.Tensor] = None):
        a, s2p1, sigma, dt, sigma_up = self.step_coeffs(timestep)
        s = sample.float()
        x0 = model_output * a + (s / s2p1)
        derivative = (s - x0) / sigma
        prev = s + derivative * dt
        if self.ancestral:
            prev = prev + noise * sigma_up
        return prev


class EulerAncestralDiscreteScheduler(EulerDiscreteScheduler):
    """diffusers `EulerAncestralDiscreteScheduler` ("Euler A"): the Euler step to sigma_down followed by fresh noise of scale
    sigma_up (sigma_up^2 = sigma_to^2 (sigma^2 - sigma_to^2) / sigma^2, sigma_down^2 = sigma_to^2 - sigma_up^2); the noise is
    `randn_tensor(model_output.shape, dtype=model_output.dtype, device=..., generator=generator)`, one draw per step."""
    ancestral = True


class DPMSolverMultistepScheduler(_SigmaScheduler):
    """diffusers `DPMSolverMultistepScheduler` ("DPM++"): algorithm_type dpmsolver++, solver_order 2, midpoint, lower_order_final,
    final sigma 0, init_noise_sigma 1, int64 timesteps.  With alpha = 1 / sqrt(sigma^2 + 1), sig = sigma alpha, lambda = log alpha - log sig:

        x0_i = alpha_i x - sig_i v                      (alpha_i x is a 0-dim fp32 scalar times the bf16 latents: rounded to bf16)
        h = lambda_{i+1} - lambda_i;   A = sig_{i+1} / sig_i;   B = alpha_{i+1} (exp(-h) - 1)
        first order:   x_next = A x - B x0_i            (first step; last step because the final sigma is 0)
        second order:  r0 = (lambda_i - lambda_{i-1}) / h;  D1 = (1 / r0)(x0_i - x0_{i-1});  x_next = A x - B x0_i - 0.5 B D1"""

    init_noise_sigma = 1.0
    solver_order = 2

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = self._grid(num_inference_steps).astype(np.int64)
        sig = np.concatenate([self._interp_sigmas(ts), [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def step_coeffs(self, timestep):
        """(alpha_s0, sig_s0, A, B, 1 / r0 or None) as 0-dim fp32 tensors; second order iff 1 / r0 is given."""
        i = self.index_for(timestep)
        n = len(self.timesteps)
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i + 1])
        alpha_s0, sigma_s0 = self._alpha_sigma(self.sigmas[i])
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        A = sigma_t / sigma_s0
        B = alpha_t * (torch.exp(-h) - 1.0)
        lower_order_final = i == n - 1                       # final_sigmas_type == "zero" (or fewer than 15 steps)
        lower_order_second = i == n - 2 and n < 15           # never taken by a second-order solver; kept for the record
        if self.lower_order_nums < 1 or lower_order_final:
            return alpha_s0, sigma_s0, A, B, None
        alpha_s1, sigma_s1 = self._alpha_sigma(self.sigmas[i - 1])
        lambda_s1 = torch.log(alpha_s1) - torch.log(sigma_s1)
        r0 = (lambda_s0 - lambda_s1) / h
        return alpha_s0, sigma_s0, A, B, 1.0 / r0

    def step(self, p: Prec, model_output: torch.Tensor, timestep, sample: torch.Tensor):
        alpha_s0, sigma_s0, A, B, inv_r0 = self.step_coeffs(timestep)
        s = sample.float()
        x0 = p.R(alpha_s0 * s) - sigma_s0 * model_output     # convert_model_output sees the un-upcast (bf16) sample
        self.model_outputs = self.model_outputs[1:] + [x0]
        if inv_r0 is None:
            prev = A * s - B * x0
        else:
            m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
            D1 = inv_r0 * (m0 - m1)
            prev = A * s - B * m0 - 0.5 * B * D1
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return prev


class PNDMScheduler:
    """diffusers `PNDMScheduler` ("PNDM" of demo.py:647-654) as `from_pretrained` builds it from the CogVideoX scheduler config:
    scaled-linear betas (NO zero-terminal-SNR rescale: the class has no such option and ignores the key), v_prediction, trailing
    spacing, set_alpha_to_one = True, skip_prk_steps = False (class default) -> the schedule opens with 12 Runge-Kutta evaluations
    (3 groups of 4 over the last 3 intervals ... 999 -> 939) followed by 47 fourth-order linear-multistep (PLMS) steps: 59 model
    evaluations for num_inference_steps = 50; `timesteps` is that 59-long list and the pipeline loops over it.  Restated from the
    published algorithm (parity unpinned).  State: `ets` (up to 4 past model outputs, fp32), the running Runge-Kutta sum, the sample at
    the start of a Runge-Kutta group.

        eps   = sqrt(a_t) v + bf16r(sqrt(1 - a_t) x)                     (v_prediction inside `_get_prev_sample`; x is the un-upcast sample)
        x_prev = bf16r(sqrt(a_prev / a_t) x) - (a_prev - a_t) eps / (a_t sqrt(1 - a_prev) + sqrt(a_t (1 - a_t) a_prev))
    """
    order = 1
    init_noise_sigma = 1.0
    pndm_order = 4

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="v_prediction", timestep_spacing="trailing", set_alpha_to_one=True, steps_offset=0,
                 skip_prk_steps=False, **ignored):
        assert beta_schedule == "scaled_linear" and prediction_type == "v_prediction" and not skip_prk_steps
        self.num_train_timesteps, self.timestep_spacing, self.steps_offset = num_train_timesteps, timestep_spacing, steps_offset
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.timesteps = None

    def scale_model_input(self, p: Prec, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        N, n = self.num_train_timesteps, num_inference_steps
        self.num_inference_steps = n
        if self.timestep_spacing == "trailing":
            ts = np.round(np.arange(N, 0, -N / n))[::-1].astype(np.int64) - 1
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, n) * (N // n)).round().astype(np.int64) + self.steps_offset
        else:
            raise ValueError(self.timestep_spacing)
        self._timesteps = ts
        prk = np.array(ts[-self.pndm_order:]).repeat(2) + np.tile(np.array([0, N // n // 2]), self.pndm_order)
        self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
        self.plms_timesteps = ts[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self.ets, self.counter, self.cur_model_output, self.cur_sample = [], 0, 0, None

    def prev_coeffs(self, timestep: int, prev_timestep: int):
        """(sqrt(a_t), sqrt(1 - a_t), sample_coeff, a_prev - a_t, denominator) as 0-dim fp32 tensors, the library's expressions."""
        a = self.alphas_cumprod[timestep]
        ap = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b, bp = 1 - a, 1 - ap
        return a ** 0.5, b ** 0.5, (ap / a) ** 0.5, ap - a, a * bp ** 0.5 + (a * b * ap) ** 0.5

    def _get_prev_sample(self, p: Prec, sample, timestep, prev_timestep, model_output):
        sa, sb, sc, diff, denom = self.prev_coeffs(int(timestep), int(prev_timestep))
        s = sample.float()
        model_output = sa * model_output + p.R(sb * s)
        return p.R(sc * s) - diff * model_output / denom

    def step(self, p: Prec, model_output: torch.Tensor, timestep, sample: torch.Tensor):
        N, n = self.num_train_timesteps, self.num_inference_steps
        timestep = int(timestep)
        if self.counter < len(self.prk_timesteps):                       # step_prk
            diff_to_prev = 0 if self.counter % 2 else N // n // 2
            prev_timestep = timestep - diff_to_prev
            timestep = int(self.prk_timesteps[self.counter // 4 * 4])
            if self.counter % 4 == 0:
                self.cur_model_output = self.cur_model_output + 1 / 6 * model_output
                self.ets.append(model_output)
                self.cur_sample = sample
            elif (self.counter - 1) % 4 == 0 or (self.counter - 2) % 4 == 0:
                self.cur_model_output = self.cur_model_output + 1 / 3 * model_output
            else:
                model_output = self.cur_model_output + 1 / 6 * model_output
                self.cur_model_output = 0
            prev = self._get_prev_sample(p, self.cur_sample, timestep, prev_timestep, model_output)
        else:                                                            # step_plms (with the three Runge-Kutta outputs in `ets`)
            assert len(self.ets) >= 3
            prev_timestep = timestep - N // n
            self.ets = self.ets[-3:] + [model_output]
            e = self.ets
            model_output = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
            prev = self._get_prev_sample(p, sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return prev
