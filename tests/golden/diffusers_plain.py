"""Plain-torch nn.Module stand-ins for the `diffusers` classes the reference's three modules import.

FIXTURE SCAFFOLDING, used ONLY by tests/golden/make_golden.py (build container).  `diffusers`
(requirements.txt:26, `diffusers>=0.30.1`) is not installed and not vendored, so the reference's
modules cannot be imported without something standing behind these names.  make_golden.py's first
generation of stand-ins wraps the ORACLE's functions (fp32 only); the modules below are a second,
independent restatement of the same published algorithms written the way the library writes them:
ordinary nn.Modules built from torch ops, so that

  * `model.to(torch.bfloat16)` runs the reference's classes in *eager bf16* with one rounding per
    torch op, exactly as the reference executes on its own hardware (the thing
    `oracle.prec.Prec("bf16_ref")` emulates), and
  * the fp32 fixtures at the product's default widths do not route through oracle/ at all.

Every class cites the reference call site that fixes its constructor arguments and call signature.
The arithmetic itself remains "restated from the published algorithm" (parity unpinned against the
library; analytic KATs in tests/test_oracle_kat.py cover the oracle's copy of the same formulas).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------- embeddings (crosstransformer3d.py:531-534,724-732)
class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels, self.flip, self.shift, self.scale = num_channels, flip_sin_to_cos, downscale_freq_shift, scale

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.shift)
        emb = torch.exp(exponent)
        emb = timesteps[:, None].float() * emb[None, :]
        emb = self.scale * emb
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        if self.num_channels % 2 == 1:
            emb = F.pad(emb, (0, 1, 0, 0))
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        assert act_fn == "silu"
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


# ---------------------------------------------------------------- normalisation (crosstransformer3d.py:195-197,211-213,556-562)
class CogVideoXLayerNormZero(nn.Module):
    def __init__(self, conditioning_dim, embedding_dim, elementwise_affine=True, eps=1e-5, bias=True):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_dim, 6 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, eps=eps, elementwise_affine=elementwise_affine)

    def forward(self, hidden_states, encoder_hidden_states, temb):
        shift, scale, gate, enc_shift, enc_scale, enc_gate = self.linear(self.silu(temb)).chunk(6, dim=1)
        hidden_states = self.norm(hidden_states) * (1 + scale)[:, None, :] + shift[:, None, :]
        encoder_hidden_states = self.norm(encoder_hidden_states) * (1 + enc_scale)[:, None, :] + enc_shift[:, None, :]
        return hidden_states, encoder_hidden_states, gate[:, None, :], enc_gate[:, None, :]


class AdaLayerNorm(nn.Module):
    def __init__(self, embedding_dim, num_embeddings=None, output_dim=None, norm_elementwise_affine=False, norm_eps=1e-5,
                 chunk_dim=0):
        super().__init__()
        assert num_embeddings is None and chunk_dim == 1                  # the only form the reference builds (:556-562)
        self.chunk_dim = chunk_dim
        output_dim = output_dim or embedding_dim * 2
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, output_dim)
        self.norm = nn.LayerNorm(output_dim // 2, norm_eps, norm_elementwise_affine)

    def forward(self, x, timestep=None, temb=None):
        temb = self.linear(self.silu(temb))
        shift, scale = temb.chunk(2, dim=1)                               # chunk_dim == 1: shift FIRST
        shift = shift[:, None, :]
        scale = scale[:, None, :]
        return self.norm(x) * (1 + scale) + shift


# ---------------------------------------------------------------- attention (crosstransformer3d.py:199-208,239-243)
def apply_rotary_emb(x, freqs_cis):
    cos, sin = freqs_cis                                                  # use_real=True, use_real_unbind_dim=-1
    cos = cos[None, None].to(x.device)
    sin = sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rotated = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rotated.float() * sin).to(x.dtype)


class CogVideoXAttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None):
        text_seq_length = encoder_hidden_states.size(1)
        hidden_states = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        batch_size = hidden_states.shape[0]
        query = attn.to_q(hidden_states)
        key = attn.to_k(hidden_states)
        value = attn.to_v(hidden_states)
        inner_dim = key.shape[-1]
        head_dim = inner_dim // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        if attn.norm_q is not None:
            query = attn.norm_q(query)
        if attn.norm_k is not None:
            key = attn.norm_k(key)
        if image_rotary_emb is not None:
            query[:, :, text_seq_length:] = apply_rotary_emb(query[:, :, text_seq_length:], image_rotary_emb)
            key[:, :, text_seq_length:] = apply_rotary_emb(key[:, :, text_seq_length:], image_rotary_emb)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0,
                                                       is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        encoder_hidden_states, hidden_states = hidden_states.split(
            [text_seq_length, hidden_states.size(1) - text_seq_length], dim=1)
        return hidden_states, encoder_hidden_states


FusedCogVideoXAttnProcessor2_0 = CogVideoXAttnProcessor2_0


class Attention(nn.Module):
    def __init__(self, query_dim, dim_head=64, heads=8, qk_norm=None, eps=1e-5, bias=False, out_bias=True, processor=None):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        if qk_norm == "layer_norm":
            self.norm_q = nn.LayerNorm(dim_head, eps=eps)
            self.norm_k = nn.LayerNorm(dim_head, eps=eps)
        else:
            assert qk_norm is None
            self.norm_q = self.norm_k = None
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])
        self.processor = processor if processor is not None else CogVideoXAttnProcessor2_0()

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, hidden_states):
        return F.gelu(self.proj(hidden_states), approximate=self.approximate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False, inner_dim=None,
                 bias=True):
        super().__init__()
        assert activation_fn == "gelu-approximate"                       # the reference's default (:182)
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        dim_out = dim_out if dim_out is not None else dim
        self.net = nn.ModuleList([GELU(dim, inner_dim, approximate="tanh", bias=bias), nn.Dropout(dropout),
                                  nn.Linear(inner_dim, dim_out, bias=bias)])
        if final_dropout:
            self.net.append(nn.Dropout(dropout))

    def forward(self, hidden_states):
        for module in self.net:
            hidden_states = module(hidden_states)
        return hidden_states


# ---------------------------------------------------------------- VAE resamplers (autoencoder_magvit.py:428,623)
class CogVideoXUpsample3D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.compress_time = compress_time

    def forward(self, inputs):
        if self.compress_time:
            if inputs.shape[2] > 1 and inputs.shape[2] % 2 == 1:
                x_first, x_rest = inputs[:, :, 0], inputs[:, :, 1:]
                x_first = F.interpolate(x_first, scale_factor=2.0)
                x_rest = F.interpolate(x_rest, scale_factor=2.0)
                x_first = x_first[:, :, None, :, :]
                inputs = torch.cat([x_first, x_rest], dim=2)
            elif inputs.shape[2] > 1:
                inputs = F.interpolate(inputs, scale_factor=2.0)
            else:
                inputs = inputs.squeeze(2)
                inputs = F.interpolate(inputs, scale_factor=2.0)
                inputs = inputs[:, :, None, :, :]
        else:
            b, c, t, h, w = inputs.shape
            inputs = inputs.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
            inputs = F.interpolate(inputs, scale_factor=2.0)
            inputs = inputs.reshape(b, t, c, *inputs.shape[2:]).permute(0, 2, 1, 3, 4)
        b, c, t, h, w = inputs.shape
        inputs = inputs.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        inputs = self.conv(inputs)
        inputs = inputs.reshape(b, t, *inputs.shape[1:]).permute(0, 2, 1, 3, 4)
        return inputs


class CogVideoXDownsample3D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=2, padding=0, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.compress_time = compress_time

    def forward(self, x):
        if self.compress_time:
            batch_size, channels, frames, height, width = x.shape
            x = x.permute(0, 3, 4, 1, 2).reshape(batch_size * height * width, channels, frames)
            if x.shape[-1] % 2 == 1:
                x_first, x_rest = x[..., 0], x[..., 1:]
                if x_rest.shape[-1] > 0:
                    x_rest = F.avg_pool1d(x_rest, kernel_size=2, stride=2)
                x = torch.cat([x_first[..., None], x_rest], dim=-1)
            else:
                x = F.avg_pool1d(x, kernel_size=2, stride=2)
            x = x.reshape(batch_size, height, width, channels, x.shape[-1]).permute(0, 3, 4, 1, 2)
        x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        batch_size, channels, frames, height, width = x.shape
        x = x.permute(0, 2, 1, 3, 4).reshape(batch_size * frames, channels, height, width)
        x = self.conv(x)
        x = x.reshape(batch_size, frames, x.shape[1], x.shape[2], x.shape[3]).permute(0, 2, 1, 3, 4)
        return x


class DiagonalGaussianDistribution:
    """autoencoder_magvit.py:1197,1212: mean | logvar = chunk(parameters, 2, dim=1), logvar clamped to [-30, 20]."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        # the library draws in parameters.dtype; a bf16 draw consumes the generator differently from an fp32 one, so the fixture
        # draws fp32 and rounds: the fp32 and the bf16 run of one seed then share the noise VALUES (the comparison is about arithmetic)
        noise = torch.randn(self.mean.shape, generator=generator, dtype=torch.float32).to(self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


def get_activation(name):
    assert name in ("silu", "swish")
    return nn.SiLU()


# ---------------------------------------------------------------- scheduler (pipeline_trajectorycrafter.py:846,1099,1165)
class DDIMSchedulerPlain:
    """`DDIMScheduler.step` written as the library's torch expressions on (fp32 model_output, activation-dtype sample), so the
    eager type promotion happens by itself (a 0-dim fp32 coefficient times a bf16 tensor stays bf16).  The schedule TABLES
    (precision-independent host arithmetic) are taken from the table builder handed in by make_golden.py."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, tables):
        self.s = tables

    def set_timesteps(self, n, device=None):
        self.s.set_timesteps(n)
        self.timesteps = self.s.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        assert eta == 0.0
        alpha_prod_t, alpha_prod_t_prev = self.s.coeffs(int(timestep))
        beta_prod_t = 1 - alpha_prod_t
        pred_original_sample = (alpha_prod_t ** 0.5) * sample - (beta_prod_t ** 0.5) * model_output      # v_prediction
        pred_epsilon = (alpha_prod_t ** 0.5) * model_output + (beta_prod_t ** 0.5) * sample
        variance = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * (1 - alpha_prod_t / alpha_prod_t_prev)
        std_dev_t = eta * variance ** 0.5
        pred_sample_direction = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5 * pred_epsilon
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        return (prev_sample,)
