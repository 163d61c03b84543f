"""Parameter inventories (reference state-dict key names) and the random-init recipe.

No checkpoints exist offline, so every test and benchmark runs on random weights of the
reference's architecture.  Key names follow the reference's module attribute paths
(models/crosstransformer3d.py:510-579, models/autoencoder_magvit.py:850-913,1027-1048) so a
real checkpoint's state dict loads unchanged.

Recipe (keeps activations O(1) through 42 blocks while exercising every gated path):
  * matrices / conv kernels:  N(0, gain / sqrt(fan_in)), gain 1 (0.5 for the AdaLN `*.linear`)
  * biases:                   N(0, 0.02)
  * LayerNorm/GroupNorm affine: weight 1 + 0.1 N(0,1), bias 0.05 N(0,1)
  * SpatialNorm conv_y bias:  1 + 0.1 N(0,1)  (it multiplies the normalised features)
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

TRANSFORMER_5B = dict(
    num_attention_heads=48, attention_head_dim=64, in_channels=33, out_channels=16, num_layers=42,
    use_rotary_positional_embeddings=True, is_train_cross=True, cross_attn_in_channels=16,
    cross_attn_interval=2, cross_attn_dim_head=128, cross_attn_num_heads=16,
)


def transformer_param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    H, dh = cfg["num_attention_heads"], cfg["attention_head_dim"]
    D = H * dh
    p = cfg.get("patch_size", 2)
    te = cfg.get("time_embed_dim", 512)
    txt = cfg.get("text_embed_dim", 4096)
    affine = cfg.get("norm_elementwise_affine", True)
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["patch_embed.proj.weight"] = (D, cfg["in_channels"], p, p)
    s["patch_embed.proj.bias"] = (D,)
    s["patch_embed.text_proj.weight"] = (D, txt)
    s["patch_embed.text_proj.bias"] = (D,)
    s["time_embedding.linear_1.weight"] = (te, D)
    s["time_embedding.linear_1.bias"] = (te,)
    s["time_embedding.linear_2.weight"] = (te, te)
    s["time_embedding.linear_2.bias"] = (te,)
    for i in range(cfg["num_layers"]):
        b = f"transformer_blocks.{i}."
        for n in ("norm1", "norm2"):
            s[b + n + ".linear.weight"] = (6 * D, te)
            s[b + n + ".linear.bias"] = (6 * D,)
            if affine:
                s[b + n + ".norm.weight"] = (D,)
                s[b + n + ".norm.bias"] = (D,)
        for n in ("to_q", "to_k", "to_v"):
            s[b + f"attn1.{n}.weight"] = (D, D)
            if cfg.get("attention_bias", True):
                s[b + f"attn1.{n}.bias"] = (D,)
        for n in ("norm_q", "norm_k"):
            s[b + f"attn1.{n}.weight"] = (dh,)
            s[b + f"attn1.{n}.bias"] = (dh,)
        s[b + "attn1.to_out.0.weight"] = (D, D)
        s[b + "attn1.to_out.0.bias"] = (D,)
        s[b + "ff.net.0.proj.weight"] = (4 * D, D)
        s[b + "ff.net.0.proj.bias"] = (4 * D,)
        s[b + "ff.net.2.weight"] = (D, 4 * D)
        s[b + "ff.net.2.bias"] = (D,)
    if affine:
        s["norm_final.weight"] = (D,)
        s["norm_final.bias"] = (D,)
    s["norm_out.linear.weight"] = (2 * D, te)
    s["norm_out.linear.bias"] = (2 * D,)
    if affine:
        s["norm_out.norm.weight"] = (D,)
        s["norm_out.norm.bias"] = (D,)
    s["proj_out.weight"] = (p * p * cfg.get("out_channels", 16), D)
    s["proj_out.bias"] = (p * p * cfg.get("out_channels", 16),)
    if cfg.get("is_train_cross", False):
        s["ref_patch_embed.proj.weight"] = (D, cfg.get("cross_attn_in_channels", 16), p, p)
        s["ref_patch_embed.proj.bias"] = (D,)
        inner = cfg.get("cross_attn_dim_head", 128) * cfg.get("cross_attn_num_heads", 16)
        for j in range(cfg["num_layers"] // cfg.get("cross_attn_interval", 2)):
            b = f"perceiver_cross_attention.{j}."
            for n in ("norm1", "norm2"):
                s[b + n + ".weight"] = (D,)
                s[b + n + ".bias"] = (D,)
            s[b + "to_q.weight"] = (inner, D)
            s[b + "to_kv.weight"] = (2 * inner, D)
            s[b + "to_out.weight"] = (D, inner)
    return s


def _resnet_shapes(s, prefix, cin, cout, zq):
    for n, c in (("norm1", cin), ("norm2", cout)):
        if zq is None:
            s[prefix + n + ".weight"] = (c,)
            s[prefix + n + ".bias"] = (c,)
        else:
            s[prefix + n + ".norm_layer.weight"] = (c,)
            s[prefix + n + ".norm_layer.bias"] = (c,)
            for cv in ("conv_y", "conv_b"):
                s[prefix + n + f".{cv}.conv.weight"] = (c, zq, 1, 1, 1)
                s[prefix + n + f".{cv}.conv.bias"] = (c,)
    s[prefix + "conv1.conv.weight"] = (cout, cin, 3, 3, 3)
    s[prefix + "conv1.conv.bias"] = (cout,)
    s[prefix + "conv2.conv.weight"] = (cout, cout, 3, 3, 3)
    s[prefix + "conv2.conv.bias"] = (cout,)
    if cin != cout:
        s[prefix + "conv_shortcut.weight"] = (cout, cin, 1, 1, 1)
        s[prefix + "conv_shortcut.bias"] = (cout,)


def vae_param_shapes(cfg: dict, decoder: bool = True, encoder: bool = True):
    boc = list(cfg.get("block_out_channels", (128, 256, 256, 512)))
    lat = cfg.get("latent_channels", 16)
    lpb = cfg.get("layers_per_block", 3)
    cin_img, cout_img = cfg.get("in_channels", 3), cfg.get("out_channels", 3)
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if encoder:
        s["encoder.conv_in.conv.weight"] = (boc[0], cin_img, 3, 3, 3)
        s["encoder.conv_in.conv.bias"] = (boc[0],)
        out_c = boc[0]
        for i, c in enumerate(boc):
            in_c, out_c = out_c, c
            for j in range(lpb):
                _resnet_shapes(s, f"encoder.down_blocks.{i}.resnets.{j}.", in_c if j == 0 else out_c, out_c, None)
            if i != len(boc) - 1:
                s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
                s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (out_c,)
        for j in range(2):
            _resnet_shapes(s, f"encoder.mid_block.resnets.{j}.", boc[-1], boc[-1], None)
        s["encoder.norm_out.weight"] = (boc[-1],)
        s["encoder.norm_out.bias"] = (boc[-1],)
        s["encoder.conv_out.conv.weight"] = (2 * lat, boc[-1], 3, 3, 3)
        s["encoder.conv_out.conv.bias"] = (2 * lat,)
    if decoder:
        rb = list(reversed(boc))
        s["decoder.conv_in.conv.weight"] = (rb[0], lat, 3, 3, 3)
        s["decoder.conv_in.conv.bias"] = (rb[0],)
        for j in range(2):
            _resnet_shapes(s, f"decoder.mid_block.resnets.{j}.", rb[0], rb[0], lat)
        out_c = rb[0]
        for i, c in enumerate(rb):
            in_c, out_c = out_c, c
            for j in range(lpb + 1):
                _resnet_shapes(s, f"decoder.up_blocks.{i}.resnets.{j}.", in_c if j == 0 else out_c, out_c, lat)
            if i != len(rb) - 1:
                s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
                s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (out_c,)
        for n in ("norm_layer",):
            s[f"decoder.norm_out.{n}.weight"] = (rb[-1],)
            s[f"decoder.norm_out.{n}.bias"] = (rb[-1],)
        for cv in ("conv_y", "conv_b"):
            s[f"decoder.norm_out.{cv}.conv.weight"] = (rb[-1], lat, 1, 1, 1)
            s[f"decoder.norm_out.{cv}.conv.bias"] = (rb[-1],)
        s["decoder.conv_out.conv.weight"] = (cout_img, rb[-1], 3, 3, 3)
        s["decoder.conv_out.conv.bias"] = (cout_img,)
    return s


def _is_norm_affine(name: str) -> bool:
    leaf = name.rsplit(".", 2)
    mod = leaf[-2] if len(leaf) >= 2 else ""
    return mod in ("norm", "norm_q", "norm_k", "norm_final", "norm_layer", "norm1", "norm2", "norm_out") \
        and not name.endswith("linear.weight") and not name.endswith("linear.bias")


def random_state_dict(shapes: "Dict[str, Tuple[int, ...]]", seed: int = 0, dtype=torch.float32,
                      device="cpu") -> "OrderedDict[str, torch.Tensor]":
    """Deterministic random init (recipe in the module docstring)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in shapes.items():
        n = torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        is_w = name.endswith("weight")
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            gain = 0.5 if ".linear." in name and ("norm1" in name or "norm2" in name or "norm_out" in name) else 1.0
            if "conv_y" in name or "conv_b" in name:
                gain = 0.3
            t = n * (gain / fan_in ** 0.5)
        elif _is_norm_affine(name):
            t = 1.0 + 0.1 * n if is_w else 0.05 * n
        elif "conv_y.conv.bias" in name:
            t = 1.0 + 0.1 * n
        else:
            t = 0.02 * n
        sd[name] = t.to(dtype)
    return sd


# ---------------------------------------------------------------------------------------------------------------
# Host-independent random init.  `random_state_dict` draws from torch's CPU generator, whose normal transform goes
# through libm / vectorised log-cos kernels: good within one host, not promised bit-identical across hosts.  The
# reference-run fixtures at the product's default widths (tests/golden/make_golden.py default) store inputs and
# outputs only and REGENERATE the ~0.5 B weights wherever the test runs, so they need a stream that is bit-exact
# on every machine: integer hashing (splitmix64) + an Irwin-Hall sum of four 16-bit fields, i.e. integer arithmetic
# and ONE float32 multiply per element.
# ---------------------------------------------------------------------------------------------------------------
_IH4_STD = 65536.0 * (1.0 / 3.0) ** 0.5            # std of the sum of four uniform 16-bit integers (to 8e-11 relative)


def hashed_normal(shape, seed: int, stream: int, scale: float = 1.0) -> torch.Tensor:
    """~N(0, scale^2) float32 tensor (Irwin-Hall n = 4: support +-3.46 sigma), bit-identical on every host."""
    n = int(np.prod(shape)) if len(shape) else 1
    out = torch.empty(n, dtype=torch.float32)
    i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v                # two's-complement view of a 64-bit constant
    lsr = lambda z, k: (z >> k) & ((1 << (64 - k)) - 1)                   # logical shift on int64 (wrap-around arithmetic = mod 2^64)
    golden, c1, c2 = i64(0x9E3779B97F4A7C15), i64(0xBF58476D1CE4E5B9), i64(0x94D049BB133111EB)
    base = i64(((((seed << 40) ^ (stream << 20)) * 0xD1B54A32D192ED03) + 0x9E3779B97F4A7C15) & ((1 << 64) - 1))
    mul = float(np.float32(float(scale) / _IH4_STD))
    step = 1 << 24
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        z = torch.arange(lo, hi, dtype=torch.int64) * golden + base
        z = (z ^ lsr(z, 30)) * c1
        z = (z ^ lsr(z, 27)) * c2
        z = z ^ lsr(z, 31)
        s = (z & 0xFFFF) + (lsr(z, 16) & 0xFFFF) + (lsr(z, 32) & 0xFFFF) + lsr(z, 48)
        out[lo:hi] = (s - 131070).to(torch.float32) * mul
    return out.reshape(tuple(shape))


def hashed_state_dict(shapes: "Dict[str, Tuple[int, ...]]", seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """The recipe of `random_state_dict` on the host-independent stream, values rounded to bf16-representable float32
    (the fixtures' models hold exactly these numbers in fp32 and in bf16)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for stream, (name, shape) in enumerate(shapes.items()):
        is_w = name.endswith("weight")
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            gain = 0.5 if ".linear." in name and ("norm1" in name or "norm2" in name or "norm_out" in name) else 1.0
            if "conv_y" in name or "conv_b" in name:
                gain = 0.3
            t = hashed_normal(shape, seed, stream, gain / fan_in ** 0.5)
        elif _is_norm_affine(name):
            t = hashed_normal(shape, seed, stream, 0.1) + 1.0 if is_w else hashed_normal(shape, seed, stream, 0.05)
        elif "conv_y.conv.bias" in name:
            t = hashed_normal(shape, seed, stream, 0.1) + 1.0
        else:
            t = hashed_normal(shape, seed, stream, 0.02)
        sd[name] = t.to(torch.bfloat16).float()
    return sd


def state_dict_digest(sd) -> str:
    """sha256 over names, shapes and the bf16 bit patterns: what a fixture stores instead of the weights."""
    import hashlib
    h = hashlib.sha256()
    for name, t in sd.items():
        h.update(name.encode())
        h.update(str(tuple(t.shape)).encode())
        h.update(t.detach().to(torch.bfloat16).contiguous().view(torch.int16).numpy().tobytes())
    return h.hexdigest()
