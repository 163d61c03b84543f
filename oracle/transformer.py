"""Oracle for `CrossTransformer3DModel.forward` (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference models/crosstransformer3d.py:47-136 (patch embeds), :224-266
(CogVideoXBlock), :376-398 (PerceiverCrossAttention), :711-871 (forward), with the
diffusers-resident pieces taken from oracle/diffusers_restated.py.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import diffusers_restated as dr
from .prec import Prec

# class defaults, reference models/crosstransformer3d.py:459-492
DEFAULT_CONFIG = dict(
    num_attention_heads=30, attention_head_dim=64, in_channels=16, out_channels=16,
    flip_sin_to_cos=True, freq_shift=0, time_embed_dim=512, text_embed_dim=4096, num_layers=30,
    dropout=0.0, attention_bias=True, sample_width=90, sample_height=60, sample_frames=49,
    patch_size=2, temporal_compression_ratio=4, max_text_seq_length=226,
    activation_fn="gelu-approximate", timestep_activation_fn="silu",
    norm_elementwise_affine=True, norm_eps=1e-5, spatial_interpolation_scale=1.875,
    temporal_interpolation_scale=1.0, use_rotary_positional_embeddings=False,
    add_noise_in_inpaint_model=False, is_train_cross=False, cross_attn_in_channels=16,
    cross_attn_interval=2, cross_attn_dim_head=128, cross_attn_num_heads=16,
)


def patch_embed_video(p: Prec, w, b, x: torch.Tensor, patch: int) -> torch.Tensor:
    """Conv2d(k=patch, s=patch) per frame, flatten (f, h', w') — reference :78-87 / :120-135."""
    B, Fr, C, H, W = x.shape
    y = x.reshape(B * Fr, C, H, W)
    y = p.R(F.conv2d(y.float(), p.param(w), p.param(b), stride=patch))
    y = y.view(B, Fr, *y.shape[1:]).flatten(3).transpose(2, 3).flatten(1, 2)
    return y


def perceiver_cross_attention(p: Prec, sd: dict, prefix: str, x, latents, heads: int, dim_head: int):
    """reference :376-398.  LN (default eps 1e-5), to_q / to_kv (no bias), q*s and k*s with
    s = dim_head**-0.25 rounded before QK^T, softmax fp32 (autocast), P rounded before PV."""
    x = p.layer_norm(x, sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"], 1e-5, contract_point=True)
    lat = p.layer_norm(latents, sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], 1e-5, contract_point=True)
    B, S, _ = lat.shape
    q = p.linear(lat, sd[prefix + "to_q.weight"])
    kv = p.linear(x, sd[prefix + "to_kv.weight"])
    k, v = kv.chunk(2, dim=-1)

    def split(t):
        return t.view(B, t.shape[1], heads, -1).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    s = 1.0 / (dim_head ** 0.25)
    if p.mode == "bf16":                                            # HIP contract: q carries log2(e) too, base-2 softmax
        qs, ks = p.R(q * (s * dr.LOG2E)), p.R(k * s)
        o = p.R(dr.sdpa_log2(p, qs, ks, v))
        o = o.permute(0, 2, 1, 3).reshape(B, S, -1)
        return p.linear_fused(o, sd[prefix + "to_out.weight"])      # consumed by the caller's residual add
    qs, ks = p.R(q * s), p.R(k * s)
    if p.mode == "bf16_ref":
        w = p.r(torch.matmul(qs, ks.transpose(-1, -2)))          # the reference materialises bf16 scores (:392)
        o = torch.matmul(p.R(torch.softmax(w, dim=-1)), v)       # :394-395
    else:
        o = dr.sdpa(p, qs, ks, v, 1.0)                           # flash contract, scores never leave fp32
    o = p.R(o)
    o = o.permute(0, 2, 1, 3).reshape(B, S, -1)
    return p.linear(o, sd[prefix + "to_out.weight"])


def cogvideox_block(p: Prec, sd: dict, prefix: str, hidden, encoder, temb, rotary, heads: int, eps: float):
    """reference :224-266."""
    text_len = encoder.shape[1]
    nh, ne, gate, e_gate = dr.layer_norm_zero(p, sd, prefix + "norm1.", hidden, encoder, temb, eps)
    ah, ae = dr.cogvideox_attention(p, sd, prefix + "attn1.", nh, ne, heads, rotary)
    hidden = p.R(hidden + p.r(gate * ah))
    encoder = p.R(encoder + p.r(e_gate * ae))

    nh, ne, gate, e_gate = dr.layer_norm_zero(p, sd, prefix + "norm2.", hidden, encoder, temb, eps)
    ff = dr.feed_forward(p, sd, prefix + "ff.", torch.cat([ne, nh], dim=1))
    hidden = p.R(hidden + p.r(gate * ff[:, text_len:]))
    encoder = p.R(encoder + p.r(e_gate * ff[:, :text_len]))
    return hidden, encoder


def sincos_position_table(p: Prec, cfg: dict, D: int, H: int, W: int, rows: int) -> torch.Tensor:
    """The `pos_embedding` buffer (reference :516-528: zeros for `max_text_seq_length` rows, then the 3-D sincos table of the
    CONFIGURED sample size, cast with the model, p.R) resized to the call's latent size as the forward does (:755-782): viewed
    [1, T_post, H_post, W_post, D] from row `text_seq_length` on, trilinear (align_corners False) to [T_post, H/p, W/p], cut to
    `rows`.  Like the reference this needs text_seq_length == max_text_seq_length for the view to fit."""
    patch = cfg["patch_size"]
    ph, pw = cfg["sample_height"] // patch, cfg["sample_width"] // patch
    pt = (cfg["sample_frames"] - 1) // cfg["temporal_compression_ratio"] + 1
    tab = dr.get_3d_sincos_pos_embed(D, (pw, ph), pt, cfg["spatial_interpolation_scale"], cfg["temporal_interpolation_scale"])
    tab = p.R(torch.from_numpy(tab).flatten(0, 1).float())                      # the buffer is fp32, then model.to(dtype)
    v = tab.view(1, pt, ph, pw, D).permute(0, 4, 1, 2, 3)
    v = p.R(torch.nn.functional.interpolate(v, size=[pt, H // patch, W // patch], mode="trilinear", align_corners=False))
    v = v.permute(0, 2, 3, 4, 1).reshape(1, -1, D)
    text = torch.zeros(1, cfg["max_text_seq_length"], D)
    return torch.cat([text, v], dim=1)[:, :rows]


def transformer_forward(sd: dict, config: dict, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor,
                        timestep: torch.Tensor, inpaint_latents: torch.Tensor, cross_latents: Optional[torch.Tensor],
                        image_rotary_emb: Optional[Tuple[torch.Tensor, torch.Tensor]], prec: str = "fp32",
                        num_blocks: Optional[int] = None, taps: Optional[dict] = None) -> torch.Tensor:
    """reference :711-871.  Returns `sample` [B,F,C,h,w] in the activation dtype."""
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(config)
    p = Prec(prec)
    heads = cfg["num_attention_heads"]
    D = heads * cfg["attention_head_dim"]
    patch = cfg["patch_size"]
    eps = cfg["norm_eps"]
    B, Fr, C, H, W = hidden_states.shape
    if inpaint_latents is None:
        raise ValueError("inpaint_latents is required (reference :736 concatenates it unconditionally)")

    hs = p.R(hidden_states)
    enc = p.R(encoder_hidden_states)
    inp = p.R(inpaint_latents)

    # 1. time embedding (:724-732)
    t_emb = dr.timesteps_proj(timestep, D, cfg["flip_sin_to_cos"], cfg["freq_shift"])
    t_emb = p.R(t_emb)
    emb = dr.timestep_embedding(p, sd, "time_embedding.", t_emb)

    # 2. patch embedding (:736-737, :68-92)
    x = torch.cat([hs, inp], dim=2)
    text = p.linear(enc, sd["patch_embed.text_proj.weight"], sd["patch_embed.text_proj.bias"])
    vid = patch_embed_video(p, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], x, patch)
    encoder, hidden = text, vid

    cross = None
    if cfg["is_train_cross"]:
        if cross_latents is None:
            raise ValueError("cross_latents is required when is_train_cross=True (reference :744-745)")
        cross = patch_embed_video(p, sd["ref_patch_embed.proj.weight"], sd["ref_patch_embed.proj.bias"],
                                  p.R(cross_latents), patch)
    if taps is not None:
        taps["patch_embed"] = hidden.clone()

    # 3. position embedding of the non-rotary (2B) model (:752-784); text rows get zeros
    if not cfg["use_rotary_positional_embeddings"]:
        text_len = encoder.shape[1]
        pos = sincos_position_table(p, cfg, D, H, W, text_len + H * W * Fr // patch ** 2).to(hidden.device)
        joint = p.R(torch.cat([encoder, hidden], dim=1) + pos)
        encoder, hidden = joint[:, :text_len], joint[:, text_len:]

    rotary = None
    if image_rotary_emb is not None:
        rotary = (image_rotary_emb[0].float(), image_rotary_emb[1].float())

    # 4. blocks (:794-838)
    n_layers = cfg["num_layers"] if num_blocks is None else num_blocks
    ca = 0
    for i in range(n_layers):
        hidden, encoder = cogvideox_block(p, sd, f"transformer_blocks.{i}.", hidden, encoder, emb, rotary, heads, eps)
        if cfg["is_train_cross"] and i % cfg["cross_attn_interval"] == 0:
            hidden = p.R(hidden + perceiver_cross_attention(
                p, sd, f"perceiver_cross_attention.{ca}.", cross, hidden,
                cfg["cross_attn_num_heads"], cfg["cross_attn_dim_head"]))
            ca += 1
        if taps is not None:
            taps[f"block_{i}"] = hidden.clone()

    # norm_final over cat(text, video), text rows dropped (:848-850; row-wise -> video rows only)
    hidden = p.layer_norm(hidden, sd.get("norm_final.weight"), sd.get("norm_final.bias"), eps, contract_point=True)
    # 5. final block (:856-857)
    hidden = dr.ada_layer_norm(p, sd, "norm_out.", hidden, emb, eps)
    hidden = p.linear(hidden, sd["proj_out.weight"], sd["proj_out.bias"])
    # 6. unpatchify (:863-867)
    out = hidden.reshape(B, Fr, H // patch, W // patch, C, patch, patch)
    out = out.permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    return p.out(out)
