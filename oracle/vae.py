"""Oracle for `AutoencoderKLCogVideoX` decode / encode (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference models/autoencoder_magvit.py: CogVideoXCausalConv3d :76-163,
CogVideoXSpatialNorm3D :166-212, CogVideoXResnetBlock3D :215-355, blocks :358-660,
encoder :663-800, decoder :803-953, encode/_decode :1176-1253, tiled_decode / blend_v / blend_h :1282-1392.  The conv cache is an
explicit dict (the reference hides it in module state, :134,157).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import diffusers_restated as dr
from .prec import Prec

# class defaults, reference models/autoencoder_magvit.py:991-1024
DEFAULT_CONFIG = dict(
    in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 512), latent_channels=16,
    layers_per_block=3, act_fn="silu", norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4,
    sample_height=480, sample_width=720, scaling_factor=1.15258426,
    use_quant_conv=False, use_post_quant_conv=False,
)


def causal_conv3d(p: Prec, sd: dict, prefix: str, x: torch.Tensor, cache: dict,
                  res: torch.Tensor | None = None) -> torch.Tensor:
    """reference :136-163.  `prefix` names the CogVideoXCausalConv3d module; weights at prefix+'conv.'.
    `res` is the resnet shortcut (:354): the HIP path adds it in the conv epilogue before the single
    rounding; the reference rounds the conv output first (per-op point)."""
    w, b = sd[prefix + "conv.weight"], sd[prefix + "conv.bias"]
    kt, kh, kw = w.shape[2:]
    if kt > 1:
        prev = cache.get(prefix)
        ctx = [prev] if prev is not None else [x[:, :, :1]] * (kt - 1)
        x = torch.cat(ctx + [x], dim=2)
        cache[prefix] = x[:, :, -(kt - 1):].clone()          # saved before spatial padding (:157)
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2), mode="constant", value=0)
    y = F.conv3d(x.float(), p.param(w), p.param(b))
    if res is not None:
        y = p.r(y) + res
    return p.R(y)


def spatial_norm3d(p: Prec, sd: dict, prefix: str, f, zq, groups: int, cache: dict, silu: bool = True):
    """reference :199-212 (+ the SiLU every caller applies right after, :333,347,951)."""
    T = f.shape[2]
    if T > 1 and T % 2 == 1:
        z_first = F.interpolate(zq[:, :, :1], size=(1,) + tuple(f.shape[-2:]))
        z_rest = F.interpolate(zq[:, :, 1:], size=(T - 1,) + tuple(f.shape[-2:]))
        zq = torch.cat([z_first, z_rest], dim=2)
    else:
        zq = F.interpolate(zq, size=tuple(f.shape[-3:]))
    nf = F.group_norm(f.float(), groups, p.param(sd[prefix + "norm_layer.weight"]),
                      p.param(sd[prefix + "norm_layer.bias"]), eps=1e-6)
    nf = p.r(nf)
    # contract points: the HIP path materialises conv_y(zq) / conv_b(zq) as bf16 tables (at zq's resolution)
    y = p.R(F.conv3d(zq.float(), p.param(sd[prefix + "conv_y.conv.weight"]), p.param(sd[prefix + "conv_y.conv.bias"])))
    bb = p.R(F.conv3d(zq.float(), p.param(sd[prefix + "conv_b.conv.weight"]), p.param(sd[prefix + "conv_b.conv.bias"])))
    out = p.r(p.r(nf * y) + bb)
    if silu:
        out = F.silu(out)
    return p.R(out)                                            # contract: fused GN+SpatialNorm+SiLU output


def group_norm_silu(p: Prec, sd: dict, prefix: str, x, groups: int, eps: float):
    nf = p.r(F.group_norm(x.float(), groups, p.param(sd[prefix + "weight"]), p.param(sd[prefix + "bias"]), eps=eps))
    return p.R(F.silu(nf))


def resnet_block3d(p: Prec, sd: dict, prefix: str, x, zq, groups: int, eps: float, cache: dict):
    """reference :320-355 (temb_channels = 0, dropout 0)."""
    if zq is not None:
        h = spatial_norm3d(p, sd, prefix + "norm1.", x, zq, groups, cache)
    else:
        h = group_norm_silu(p, sd, prefix + "norm1.", x, groups, eps)
    h = causal_conv3d(p, sd, prefix + "conv1.", h, cache)
    if zq is not None:
        h = spatial_norm3d(p, sd, prefix + "norm2.", h, zq, groups, cache)
    else:
        h = group_norm_silu(p, sd, prefix + "norm2.", h, groups, eps)
    if prefix + "conv_shortcut.weight" in sd:                  # CogVideoXSafeConv3d 1x1x1 (:312-318)
        x = p.R(F.conv3d(x.float(), p.param(sd[prefix + "conv_shortcut.weight"]),
                         p.param(sd[prefix + "conv_shortcut.bias"])))
    return causal_conv3d(p, sd, prefix + "conv2.", h, cache, res=x)


def decoder_forward(p: Prec, sd: dict, cfg: dict, z: torch.Tensor, cache: dict) -> torch.Tensor:
    """reference CogVideoXDecoder3D.forward :917-953 on one temporal chunk."""
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    boc = list(reversed(cfg["block_out_channels"]))
    tlevel = int(np.log2(cfg["temporal_compression_ratio"]))
    h = causal_conv3d(p, sd, "decoder.conv_in.", z, cache)
    for j in range(2):
        h = resnet_block3d(p, sd, f"decoder.mid_block.resnets.{j}.", h, z, groups, eps, cache)
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet_block3d(p, sd, f"decoder.up_blocks.{i}.resnets.{j}.", h, z, groups, eps, cache)
        if i != len(boc) - 1:
            h = dr.upsample3d(p, sd, f"decoder.up_blocks.{i}.upsamplers.0.", h, compress_time=i < tlevel)
    h = spatial_norm3d(p, sd, "decoder.norm_out.", h, z, groups, cache)
    return causal_conv3d(p, sd, "decoder.conv_out.", h, cache)


def encoder_forward(p: Prec, sd: dict, cfg: dict, x: torch.Tensor, cache: dict) -> torch.Tensor:
    """reference CogVideoXEncoder3D.forward :773-800 on one temporal chunk."""
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    boc = list(cfg["block_out_channels"])
    tlevel = int(np.log2(cfg["temporal_compression_ratio"]))
    h = causal_conv3d(p, sd, "encoder.conv_in.", x, cache)
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            h = resnet_block3d(p, sd, f"encoder.down_blocks.{i}.resnets.{j}.", h, None, groups, eps, cache)
        if i != len(boc) - 1:
            h = dr.downsample3d(p, sd, f"encoder.down_blocks.{i}.downsamplers.0.", h, compress_time=i < tlevel)
    for j in range(2):
        h = resnet_block3d(p, sd, f"encoder.mid_block.resnets.{j}.", h, None, groups, eps, cache)
    h = group_norm_silu(p, sd, "encoder.norm_out.", h, groups, 1e-6)
    return causal_conv3d(p, sd, "encoder.conv_out.", h, cache)


def _chunks(num_frames: int, fbs: int):
    """Temporal chunking shared by encode (:1199-1205) and _decode (:1235-1241)."""
    rem = num_frames % fbs
    for i in range(num_frames // fbs):
        yield fbs * i + (0 if i == 0 else rem), fbs * (i + 1) + rem


def vae_decode(sd: dict, config: dict, z: torch.Tensor, prec: str = "fp32",
               max_chunks: Optional[int] = None) -> torch.Tensor:
    """reference `_decode` :1217-1253: z [B,16,T,h,w] -> [B,3,T',8h,8w] (activation dtype)."""
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(config)
    p = Prec(prec)
    z = p.R(z)
    T = z.shape[2]
    cache: dict = {}
    if T == 1:
        return p.out(decoder_forward(p, sd, cfg, z, cache))
    dec = []
    for n, (s, e) in enumerate(_chunks(T, 2)):
        if max_chunks is not None and n >= max_chunks:
            break
        dec.append(decoder_forward(p, sd, cfg, z[:, :, s:e], cache))
    return p.out(torch.cat(dec, dim=2))


def tiling_params(config: dict, tile_sample_min_height=None, tile_sample_min_width=None,
                  tile_overlap_factor_height=None, tile_overlap_factor_width=None) -> dict:
    """The numbers `__init__` (:1081-1098) and `enable_tiling` (:1109-1153) derive (`x or default`, as the reference)."""
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(config)
    down = 2 ** (len(cfg["block_out_channels"]) - 1)
    th = tile_sample_min_height or cfg["sample_height"] // 2
    tw = tile_sample_min_width or cfg["sample_width"] // 2
    return dict(tile_sample_min_height=th, tile_sample_min_width=tw,
                tile_latent_min_height=int(th / down), tile_latent_min_width=int(tw / down),
                tile_overlap_factor_height=tile_overlap_factor_height or 1 / 6,
                tile_overlap_factor_width=tile_overlap_factor_width or 1 / 5)


def _blend(p: Prec, a: torch.Tensor, b: torch.Tensor, blend_extent: int, dim: int) -> torch.Tensor:
    """blend_v (dim 3, :1282-1291) / blend_h (dim 4, :1293-1301): IN PLACE on b; in bf16 the reference's eager arithmetic rounds
    each product and the sum (python-float weights enter at fp32)."""
    blend_extent = min(a.shape[dim], b.shape[dim], blend_extent)
    for y in range(blend_extent):
        wa = float(np.float32(1 - y / blend_extent))
        wb = float(np.float32(y / blend_extent))
        ia = [slice(None)] * 5
        ib = [slice(None)] * 5
        ia[dim] = a.shape[dim] - blend_extent + y
        ib[dim] = y
        b[tuple(ib)] = p.R(p.R(a[tuple(ia)] * wa) + p.R(b[tuple(ib)] * wb))
    return b


def vae_tiled_decode(sd: dict, config: dict, z: torch.Tensor, prec: str = "fp32", **tiling) -> torch.Tensor:
    """reference `tiled_decode` :1303-1392 (entered from `_decode` :1222-1225 when the latent exceeds the tile): every spatial tile
    is decoded with its own conv cache over the temporal chunks, blended with the ALREADY BLENDED tile above and to its left
    (the blends write in place, :1376-1379), cropped to the row limits and concatenated."""
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(config)
    tp = tiling_params(config, **tiling)
    p = Prec(prec)
    z = p.R(z)
    T, height, width = z.shape[2:]
    lh, lw = tp["tile_latent_min_height"], tp["tile_latent_min_width"]
    if not (width > lw or height > lh):
        return vae_decode(sd, config, z, prec)
    overlap_h = int(lh * (1 - tp["tile_overlap_factor_height"]))
    overlap_w = int(lw * (1 - tp["tile_overlap_factor_width"]))
    blend_h_ext = int(tp["tile_sample_min_height"] * tp["tile_overlap_factor_height"])
    blend_w_ext = int(tp["tile_sample_min_width"] * tp["tile_overlap_factor_width"])
    limit_h = tp["tile_sample_min_height"] - blend_h_ext
    limit_w = tp["tile_sample_min_width"] - blend_w_ext
    rows = []
    for i in range(0, height, overlap_h):
        row = []
        for j in range(0, width, overlap_w):
            cache: dict = {}
            time = [decoder_forward(p, sd, cfg, z[:, :, s:e, i:i + lh, j:j + lw], cache) for s, e in _chunks(T, 2)]
            row.append(torch.cat(time, dim=2))           # T == 1: an empty list, the reference raises here too (:1345-1364)
        rows.append(row)
    result_rows = []
    for i, row in enumerate(rows):
        result_row = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = _blend(p, rows[i - 1][j], tile, blend_h_ext, 3)
            if j > 0:
                tile = _blend(p, row[j - 1], tile, blend_w_ext, 4)
            result_row.append(tile[:, :, :, :limit_h, :limit_w])
        result_rows.append(torch.cat(result_row, dim=4))
    return p.out(torch.cat(result_rows, dim=3))


def vae_encode(sd: dict, config: dict, x: torch.Tensor, prec: str = "fp32") -> dr.DiagonalGaussian:
    """reference `encode` :1176-1215: x [B,3,F,H,W] -> posterior over [B,16,T,h,w]."""
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(config)
    p = Prec(prec)
    x = p.R(x)
    Fr = x.shape[2]
    cache: dict = {}
    if Fr == 1:
        h = encoder_forward(p, sd, cfg, x, cache)
    else:
        h = torch.cat([encoder_forward(p, sd, cfg, x[:, :, s:e], cache) for s, e in _chunks(Fr, 4)], dim=2)
    return dr.DiagonalGaussian(h)
