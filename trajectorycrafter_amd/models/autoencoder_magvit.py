"""MI355X-native `AutoencoderKLCogVideoX` — drop-in for reference models/autoencoder_magvit.py.

Same class / sub-module / parameter names (state-dict keys) and the same `encode` / `decode`
surface (:1176-1280).  Internally activations are channels-last `[N, T, H, W, C]` bf16 (the
"(T*H*W, C)" layout of the north star): every convolution is the hand-written implicit-GEMM MFMA
kernel `tcx_conv3d_cl` (causal temporal context, spatial zero padding, nearest x2 upsample and the
temporal frame map are folded into its gather, so no padded / upsampled tensor is materialised),
GroupNorm + SpatialNorm + SiLU is one two-pass fused kernel pair, residual adds are conv epilogues.

Every sub-module also has the reference's own `forward(...)` signature on [N, C, T, H, W] tensors (same kernels behind a
layout conversion; tests/test_signatures.py compares the signatures with the imported reference's).  The leaf nn.Conv2d /
nn.GroupNorm modules only hold parameters; there is no torch fallback — CPU or non-bf16 calls raise `TcxError`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .._lib import TcxError
from ..config import ConfigMixin, ModelMixin, register_to_config

BF16 = torch.bfloat16


# ------------------------------------------------------------------------------ index maps
def upsample_t_map(T: int, compress_time: bool) -> List[int]:
    """Source frame of every output frame of diffusers CogVideoXUpsample3D (nearest in time)."""
    if not compress_time or T == 1:
        return list(range(T))
    if T % 2 == 1:                                   # first frame spatial-only, the rest x2 in time
        return [0] + [1 + j // 2 for j in range(2 * (T - 1))]
    return [j // 2 for j in range(2 * T)]


def zq_t_map(T: int, Tz: int) -> List[int]:
    """zq frame for every frame of f in CogVideoXSpatialNorm3D (reference :200-208)."""
    if T > 1 and T % 2 == 1:
        rest = [1 + (j * (Tz - 1)) // (T - 1) for j in range(T - 1)] if Tz > 1 else [0] * (T - 1)
        return [0] + rest
    return [(j * Tz) // T for j in range(T)]


_MAP_CACHE = {}


def _dev_map(vals: List[int], device) -> torch.Tensor:
    key = (tuple(vals), str(device))
    t = _MAP_CACHE.get(key)
    if t is None:
        t = torch.tensor(vals, dtype=torch.int32, device=device)
        _MAP_CACHE[key] = t
    return t


class _PermutedWeight:
    """Lazily cached channels-last copy [Cout, kT, kH, kW, Cin] of a conv weight."""

    def __init__(self):
        self._key = None
        self._w = None

    def get(self, w: torch.Tensor) -> torch.Tensor:
        key = (w.data_ptr(), w._version, w.dtype, w.device)
        if self._key != key:
            if w.dim() == 5:
                wp = w.detach().permute(0, 2, 3, 4, 1)
            else:                                     # Conv2d [Cout, Cin, kH, kW] -> kT = 1
                wp = w.detach().permute(0, 2, 3, 1).unsqueeze(1)
            cin = wp.shape[-1]
            if cin % 8:                               # pad input channels (RGB) to a multiple of 8 with zeros
                wp = torch.nn.functional.pad(wp, (0, 8 - cin % 8))
            self._w = wp.contiguous()
            self._key = key
        return self._w


def _pad_channels(x: torch.Tensor, mult: int = 8) -> torch.Tensor:
    c = x.shape[-1]
    return x if c % mult == 0 else torch.nn.functional.pad(x, (0, mult - c % mult))


def _to_cl(x: torch.Tensor) -> torch.Tensor:
    """The reference's [N,C,T,H,W] -> this module's channels-last [N,T,H,W,C] (tcx_ncthw_to_cl)."""
    if x.dim() != 5:
        raise ValueError(f"expected a [N, C, T, H, W] tensor, got {tuple(x.shape)}")
    return ops.ncthw_to_cl(x, 1.0)


def _from_cl(y: torch.Tensor) -> torch.Tensor:
    return y.permute(0, 4, 1, 2, 3).contiguous()


def _no_temb(temb, who: str) -> None:
    if temb is not None:
        raise NotImplementedError(f"{who}: the VAE builds its blocks with temb_channels=0 and never passes `temb` (reference :784,795,"
                                  ":934,944); a time embedding is not built on this path")


# The sub-modules below carry two entry points.  `forward_cl(...)` on channels-last activations is what the composed
# encoder / decoder run.  `forward(...)` has the REFERENCE's signature and layout ([N,C,T,H,W] in and out) and runs the same HIP
# kernels behind a layout conversion, so code written against the reference's sub-modules (`resnet(x, temb, zq)`,
# `norm(f, zq)`, `conv(x)`) works; tests/test_signatures.py compares the signatures with the reference's mechanically.
class CogVideoXSafeConv3d(nn.Conv3d):
    """reference :41-73.  The >2 GiB chunking is unnecessary (no cuDNN workspace)."""

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        """reference :61-73: the module's own convolution of an NCTHW tensor — no causal context, the module's `padding`
        spatially, none in time beyond it (a "valid" temporal convolution: the first kT-1 frames serve as context)."""
        kt, kh, kw = self.kernel_size
        if self.stride != (1, 1, 1) or self.dilation != (1, 1, 1) or self.padding[0] != 0 or self.groups != 1:
            raise NotImplementedError("CogVideoXSafeConv3d.forward: stride / dilation 1, no temporal padding, groups 1 on this path")
        if not hasattr(self, "_wcl"):
            self._wcl = _PermutedWeight()
        x = _pad_channels(_to_cl(input))
        N, T, H, W, _ = x.shape
        if T < kt:
            raise ValueError(f"CogVideoXSafeConv3d.forward: {T} frames for a temporal kernel of {kt}")
        ph, pw = self.padding[1], self.padding[2]
        cache = x[:, :kt - 1].contiguous() if kt > 1 else None
        y = ops.conv3d_cl(x[:, kt - 1:].contiguous(), self._wcl.get(self.weight), self.bias, cache=cache, pad=(ph, pw),
                          out_hw=(H + 2 * ph - kh + 1, W + 2 * pw - kw + 1))
        return _from_cl(y)

    def forward_cl(self, x: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not hasattr(self, "_wcl"):
            self._wcl = _PermutedWeight()
        return ops.conv3d_cl(x, self._wcl.get(self.weight), self.bias, res=res)


class CogVideoXCausalConv3d(nn.Module):
    """reference :76-163."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Union[int, Tuple[int, int, int]], stride: int = 1,
                 dilation: int = 1, pad_mode: str = "constant"):
        super().__init__()
        if isinstance(kernel_size, int):
            kernel_size = (kernel_size,) * 3
        if stride != 1 or dilation != 1:
            raise ValueError("only stride=1, dilation=1 causal convolutions exist on this path")
        self.time_kernel_size = kernel_size[0]
        self.conv = CogVideoXSafeConv3d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size)
        self.conv_cache: Optional[torch.Tensor] = None       # channels-last [N, kT-1, H, W, Cin]
        self._wcl = _PermutedWeight()

    def _clear_fake_context_parallel_cache(self):
        self.conv_cache = None

    def fake_context_parallel_forward(self, inputs: torch.Tensor) -> torch.Tensor:
        """reference :135-143: the NCTHW input with its causal context in front — the cached last kT-1 frames of the previous
        chunk, or frame 0 repeated.  (The HIP conv never builds this tensor: the context is address arithmetic in its gather.)"""
        kt = self.time_kernel_size
        if kt > 1:
            if self.conv_cache is not None:
                ctx = self.conv_cache[..., :inputs.shape[1]].permute(0, 4, 1, 2, 3).to(inputs.dtype)
                inputs = torch.cat([ctx, inputs], dim=2)
            else:
                inputs = torch.cat([inputs[:, :, :1]] * (kt - 1) + [inputs], dim=2)
        return inputs

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        """reference :149-163 on its NCTHW layout; updates `conv_cache` like the reference."""
        return _from_cl(self.forward_cl(_to_cl(inputs)))

    def forward_cl(self, x: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        kt = self.time_kernel_size
        x = _pad_channels(x)
        y = ops.conv3d_cl(x, self._wcl.get(self.conv.weight), self.conv.bias, cache=self.conv_cache, res=res)
        if kt > 1:                                           # cache = last kT-1 logical input frames (:157)
            if x.shape[1] >= kt - 1:
                self.conv_cache = x[:, -(kt - 1):].contiguous()
            else:
                prev = self.conv_cache if self.conv_cache is not None else x[:, :1].expand(-1, kt - 1, -1, -1, -1)
                self.conv_cache = torch.cat([prev, x], dim=1)[:, -(kt - 1):].contiguous()
        return y


class CogVideoXSpatialNorm3D(nn.Module):
    """reference :166-212 (+ the SiLU its callers apply next)."""

    def __init__(self, f_channels: int, zq_channels: int, groups: int = 32):
        super().__init__()
        self.norm_layer = nn.GroupNorm(num_channels=f_channels, num_groups=groups, eps=1e-6, affine=True)
        self.conv_y = CogVideoXCausalConv3d(zq_channels, f_channels, kernel_size=1, stride=1)
        self.conv_b = CogVideoXCausalConv3d(zq_channels, f_channels, kernel_size=1, stride=1)
        self.groups = groups

    def forward(self, f: torch.Tensor, zq: torch.Tensor) -> torch.Tensor:
        """reference :183-212 on NCTHW tensors: `norm_layer(f) * conv_y(zq') + conv_b(zq')`, zq resized to f (no activation)."""
        return _from_cl(self.forward_cl(_to_cl(f), _to_cl(zq), silu=False))

    def forward_cl(self, f: torch.Tensor, zq: torch.Tensor, silu: bool = True) -> torch.Tensor:
        T, Tz = f.shape[1], zq.shape[1]
        ytab = self.conv_y.forward_cl(zq)                     # 1x1x1 convs at zq's own resolution
        btab = self.conv_b.forward_cl(zq)
        stats = ops.groupnorm_stats(f, self.groups, self.norm_layer.eps)
        tmap = _dev_map(zq_t_map(T, Tz), f.device)
        return ops.groupnorm_apply(f, stats, self.norm_layer.weight, self.norm_layer.bias, self.groups, ytab, btab, tmap, silu)


def _groupnorm_silu(norm: nn.GroupNorm, x: torch.Tensor) -> torch.Tensor:
    stats = ops.groupnorm_stats(x, norm.num_groups, norm.eps)
    return ops.groupnorm_apply(x, stats, norm.weight, norm.bias, norm.num_groups, silu=True)


class CogVideoXResnetBlock3D(nn.Module):
    """reference :215-355 (temb_channels = 0 on this path)."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, dropout: float = 0.0, temb_channels: int = 512,
                 groups: int = 32, eps: float = 1e-6, non_linearity: str = "swish", conv_shortcut: bool = False,
                 spatial_norm_dim: Optional[int] = None, pad_mode: str = "first"):
        super().__init__()
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = conv_shortcut
        if spatial_norm_dim is None:
            self.norm1 = nn.GroupNorm(num_channels=in_channels, num_groups=groups, eps=eps)
            self.norm2 = nn.GroupNorm(num_channels=out_channels, num_groups=groups, eps=eps)
        else:
            self.norm1 = CogVideoXSpatialNorm3D(in_channels, spatial_norm_dim, groups)
            self.norm2 = CogVideoXSpatialNorm3D(out_channels, spatial_norm_dim, groups)
        self.conv1 = CogVideoXCausalConv3d(in_channels, out_channels, kernel_size=3, pad_mode=pad_mode)
        if temb_channels > 0:
            self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = CogVideoXCausalConv3d(out_channels, out_channels, kernel_size=3, pad_mode=pad_mode)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = CogVideoXCausalConv3d(in_channels, out_channels, kernel_size=3, pad_mode=pad_mode)
            else:
                self.conv_shortcut = CogVideoXSafeConv3d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, inputs: torch.Tensor, temb: Optional[torch.Tensor] = None, zq: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference :320-355 on NCTHW tensors (`zq` given for the decoder's SpatialNorm blocks, None for the encoder's)."""
        _no_temb(temb, "CogVideoXResnetBlock3D.forward")
        return _from_cl(self.forward_cl(_to_cl(inputs), None if zq is None else _to_cl(zq)))

    def forward_cl(self, x: torch.Tensor, zq: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = self.norm1.forward_cl(x, zq) if zq is not None else _groupnorm_silu(self.norm1, x)       # :328-333
        h = self.conv1.forward_cl(h)                                                                 # :334
        h = self.norm2.forward_cl(h, zq) if zq is not None else _groupnorm_silu(self.norm2, h)       # :342-347
        sc = x if self.in_channels == self.out_channels else self.conv_shortcut.forward_cl(x)        # :351-352
        return self.conv2.forward_cl(h, res=sc)                                                      # :349,354 (fused add)


class CogVideoXUpsample3D(nn.Module):
    """diffusers CogVideoXUpsample3D: nearest x2 (+ temporal rule) fused into the 3x3 conv's gather."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, padding: int = 1,
                 compress_time: bool = False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.compress_time = compress_time
        self._wcl = _PermutedWeight()

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        """diffusers `CogVideoXUpsample3D.forward` on an NCTHW tensor."""
        return _from_cl(self.forward_cl(_to_cl(inputs)))

    def forward_cl(self, x: torch.Tensor) -> torch.Tensor:
        tmap = _dev_map(upsample_t_map(x.shape[1], self.compress_time), x.device)
        return ops.conv3d_cl(x, self._wcl.get(self.conv.weight), self.conv.bias, ups=1, t_map=tmap)


class CogVideoXDownsample3D(nn.Module):
    """diffusers CogVideoXDownsample3D: temporal avg-pool (compress_time) then per-frame Conv2d 3x3 stride 2 on the
    (0,1,0,1)-padded frame — the padding is the conv gather's range check, never a tensor."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 2, padding: int = 0,
                 compress_time: bool = False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.compress_time = compress_time
        self._wcl = _PermutedWeight()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """diffusers `CogVideoXDownsample3D.forward` on an NCTHW tensor."""
        return _from_cl(self.forward_cl(_to_cl(x)))

    def forward_cl(self, x: torch.Tensor) -> torch.Tensor:
        if self.compress_time and x.shape[1] > 1:
            x = ops.avgpool_t(x)
        N, T, H, W, C = x.shape
        return ops.conv3d_cl(x, self._wcl.get(self.conv.weight), self.conv.bias, stride=2, pad=(0, 0),
                             out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1))


class CogVideoXDownBlock3D(nn.Module):
    """reference :358-464."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0, num_layers: int = 1,
                 resnet_eps: float = 1e-6, resnet_act_fn: str = "swish", resnet_groups: int = 32, add_downsample: bool = True,
                 downsample_padding: int = 0, compress_time: bool = False, pad_mode: str = "first"):
        super().__init__()
        self.resnets = nn.ModuleList([
            CogVideoXResnetBlock3D(in_channels if i == 0 else out_channels, out_channels, dropout, temb_channels,
                                   resnet_groups, resnet_eps, resnet_act_fn, pad_mode=pad_mode) for i in range(num_layers)])
        self.downsamplers = None
        if add_downsample:
            self.downsamplers = nn.ModuleList([CogVideoXDownsample3D(out_channels, out_channels, padding=downsample_padding,
                                                                     compress_time=compress_time)])

    def forward(self, hidden_states: torch.Tensor, temb: Optional[torch.Tensor] = None, zq: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference :436-464 on NCTHW tensors (`zq` is accepted and unused by the encoder's blocks, as in the reference)."""
        _no_temb(temb, "CogVideoXDownBlock3D.forward")
        return _from_cl(self.forward_cl(_to_cl(hidden_states)))

    def forward_cl(self, x):
        for r in self.resnets:
            x = r.forward_cl(x, None)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d.forward_cl(x)
        return x


class CogVideoXMidBlock3D(nn.Module):
    """reference :467-548."""

    def __init__(self, in_channels: int, temb_channels: int, dropout: float = 0.0, num_layers: int = 1, resnet_eps: float = 1e-6,
                 resnet_act_fn: str = "swish", resnet_groups: int = 32, spatial_norm_dim: Optional[int] = None,
                 pad_mode: str = "first"):
        super().__init__()
        self.resnets = nn.ModuleList([
            CogVideoXResnetBlock3D(in_channels, in_channels, dropout, temb_channels, resnet_groups, resnet_eps, resnet_act_fn,
                                   spatial_norm_dim=spatial_norm_dim, pad_mode=pad_mode) for _ in range(num_layers)])

    def forward(self, hidden_states: torch.Tensor, temb: Optional[torch.Tensor] = None, zq: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference :524-548 on NCTHW tensors."""
        _no_temb(temb, "CogVideoXMidBlock3D.forward")
        return _from_cl(self.forward_cl(_to_cl(hidden_states), None if zq is None else _to_cl(zq)))

    def forward_cl(self, x, zq=None):
        for r in self.resnets:
            x = r.forward_cl(x, zq)
        return x


class CogVideoXUpBlock3D(nn.Module):
    """reference :551-660."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0, num_layers: int = 1,
                 resnet_eps: float = 1e-6, resnet_act_fn: str = "swish", resnet_groups: int = 32, spatial_norm_dim: int = 16,
                 add_upsample: bool = True, upsample_padding: int = 1, compress_time: bool = False, pad_mode: str = "first"):
        super().__init__()
        self.resnets = nn.ModuleList([
            CogVideoXResnetBlock3D(in_channels if i == 0 else out_channels, out_channels, dropout, temb_channels,
                                   resnet_groups, resnet_eps, resnet_act_fn, spatial_norm_dim=spatial_norm_dim,
                                   pad_mode=pad_mode) for i in range(num_layers)])
        self.upsamplers = None
        if add_upsample:
            self.upsamplers = nn.ModuleList([CogVideoXUpsample3D(out_channels, out_channels, padding=upsample_padding,
                                                                 compress_time=compress_time)])

    def forward(self, hidden_states: torch.Tensor, temb: Optional[torch.Tensor] = None, zq: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference :631-660 on NCTHW tensors."""
        _no_temb(temb, "CogVideoXUpBlock3D.forward")
        return _from_cl(self.forward_cl(_to_cl(hidden_states), None if zq is None else _to_cl(zq)))

    def forward_cl(self, x, zq):
        for r in self.resnets:
            x = r.forward_cl(x, zq)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u.forward_cl(x)
        return x


class CogVideoXEncoder3D(nn.Module):
    """reference :663-800."""

    def __init__(self, in_channels: int = 3, out_channels: int = 16,
                 down_block_types: Tuple[str, ...] = ("CogVideoXDownBlock3D",) * 4,
                 block_out_channels: Tuple[int, ...] = (128, 256, 256, 512), layers_per_block: int = 3, act_fn: str = "silu",
                 norm_eps: float = 1e-6, norm_num_groups: int = 32, dropout: float = 0.0, pad_mode: str = "first",
                 temporal_compression_ratio: float = 4):
        super().__init__()
        tlevel = int(np.log2(temporal_compression_ratio))
        self.conv_in = CogVideoXCausalConv3d(in_channels, block_out_channels[0], kernel_size=3, pad_mode=pad_mode)
        self.down_blocks = nn.ModuleList([])
        output_channel = block_out_channels[0]
        for i, _ in enumerate(down_block_types):
            input_channel, output_channel = output_channel, block_out_channels[i]
            self.down_blocks.append(CogVideoXDownBlock3D(
                input_channel, output_channel, temb_channels=0, dropout=dropout, num_layers=layers_per_block,
                resnet_eps=norm_eps, resnet_act_fn=act_fn, resnet_groups=norm_num_groups,
                add_downsample=i != len(block_out_channels) - 1, compress_time=i < tlevel))
        self.mid_block = CogVideoXMidBlock3D(block_out_channels[-1], temb_channels=0, dropout=dropout, num_layers=2,
                                             resnet_eps=norm_eps, resnet_act_fn=act_fn, resnet_groups=norm_num_groups,
                                             pad_mode=pad_mode)
        self.norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = CogVideoXCausalConv3d(block_out_channels[-1], 2 * out_channels, kernel_size=3, pad_mode=pad_mode)

    def forward(self, sample: torch.Tensor, temb=None) -> torch.Tensor:
        """reference :773-800 on its NCTHW layout (one temporal chunk; the conv caches persist across calls like the reference's)."""
        return self.forward_cl(_pad_channels(ops.ncthw_to_cl(sample, 1.0))).permute(0, 4, 1, 2, 3).contiguous()

    def forward_cl(self, x: torch.Tensor) -> torch.Tensor:
        """x channels-last [N,T,H,W,3(+pad)] -> moments [N,T',H/8,W/8,32] (reference :773-800)."""
        h = self.conv_in.forward_cl(x)
        for blk in self.down_blocks:
            h = blk.forward_cl(h)
        h = self.mid_block.forward_cl(h, None)
        h = _groupnorm_silu(self.norm_out, h)
        return self.conv_out.forward_cl(h)


class CogVideoXDecoder3D(nn.Module):
    """reference :803-953."""

    def __init__(self, in_channels: int = 16, out_channels: int = 3,
                 up_block_types: Tuple[str, ...] = ("CogVideoXUpBlock3D",) * 4,
                 block_out_channels: Tuple[int, ...] = (128, 256, 256, 512), layers_per_block: int = 3, act_fn: str = "silu",
                 norm_eps: float = 1e-6, norm_num_groups: int = 32, dropout: float = 0.0, pad_mode: str = "first",
                 temporal_compression_ratio: float = 4):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = CogVideoXCausalConv3d(in_channels, rev[0], kernel_size=3, pad_mode=pad_mode)
        self.mid_block = CogVideoXMidBlock3D(rev[0], temb_channels=0, num_layers=2, resnet_eps=norm_eps, resnet_act_fn=act_fn,
                                             resnet_groups=norm_num_groups, spatial_norm_dim=in_channels, pad_mode=pad_mode)
        self.up_blocks = nn.ModuleList([])
        output_channel = rev[0]
        tlevel = int(np.log2(temporal_compression_ratio))
        for i, t in enumerate(up_block_types):
            if t != "CogVideoXUpBlock3D":
                raise ValueError("Invalid `up_block_type` encountered. Must be `CogVideoXUpBlock3D`")
            prev, output_channel = output_channel, rev[i]
            self.up_blocks.append(CogVideoXUpBlock3D(
                prev, output_channel, temb_channels=0, dropout=dropout, num_layers=layers_per_block + 1, resnet_eps=norm_eps,
                resnet_act_fn=act_fn, resnet_groups=norm_num_groups, spatial_norm_dim=in_channels,
                add_upsample=i != len(block_out_channels) - 1, compress_time=i < tlevel, pad_mode=pad_mode))
        self.norm_out = CogVideoXSpatialNorm3D(rev[-1], in_channels, groups=norm_num_groups)
        self.conv_act = nn.SiLU()
        self.conv_out = CogVideoXCausalConv3d(rev[-1], out_channels, kernel_size=3, pad_mode=pad_mode)

    def forward(self, sample: torch.Tensor, temb=None) -> torch.Tensor:
        """reference :917-953 on its NCTHW layout (one temporal chunk; the conv caches persist across calls like the reference's)."""
        return self.forward_cl(ops.ncthw_to_cl(sample, 1.0)).permute(0, 4, 1, 2, 3).contiguous()

    def forward_cl(self, z: torch.Tensor) -> torch.Tensor:
        """z channels-last [N,T,h,w,16] -> [N,T',8h,8w,3] (reference :917-953)."""
        h = self.conv_in.forward_cl(z)
        h = self.mid_block.forward_cl(h, z)
        for up in self.up_blocks:
            h = up.forward_cl(h, z)
        h = self.norm_out.forward_cl(h, z, silu=True)              # norm_out + conv_act
        return self.conv_out.forward_cl(h)


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class DiagonalGaussianDistribution:
    """diffusers DiagonalGaussianDistribution (used at reference :1197,1212)."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        dev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=dev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution

    def __getitem__(self, i):
        return (self.latent_dist,)[i]


class AutoencoderKLCogVideoX(ModelMixin, ConfigMixin):
    """reference :956-1410."""

    _supports_gradient_checkpointing = False

    @register_to_config
    def __init__(
        self,
        in_channels: int = 3,
        out_channels: int = 3,
        down_block_types: Tuple[str] = ("CogVideoXDownBlock3D",) * 4,
        up_block_types: Tuple[str] = ("CogVideoXUpBlock3D",) * 4,
        block_out_channels: Tuple[int] = (128, 256, 256, 512),
        latent_channels: int = 16,
        layers_per_block: int = 3,
        act_fn: str = "silu",
        norm_eps: float = 1e-6,
        norm_num_groups: int = 32,
        temporal_compression_ratio: float = 4,
        sample_height: int = 480,
        sample_width: int = 720,
        scaling_factor: float = 1.15258426,
        shift_factor: Optional[float] = None,
        latents_mean: Optional[Tuple[float]] = None,
        latents_std: Optional[Tuple[float]] = None,
        force_upcast: float = True,
        use_quant_conv: bool = False,
        use_post_quant_conv: bool = False,
    ):
        super().__init__()
        if use_quant_conv or use_post_quant_conv:
            raise ValueError("quant / post-quant convs are not used by CogVideoX-Fun and are not built")
        self.encoder = CogVideoXEncoder3D(in_channels, latent_channels, tuple(down_block_types), tuple(block_out_channels),
                                          layers_per_block, act_fn, norm_eps, norm_num_groups,
                                          temporal_compression_ratio=temporal_compression_ratio)
        self.decoder = CogVideoXDecoder3D(latent_channels, out_channels, tuple(up_block_types), tuple(block_out_channels),
                                          layers_per_block, act_fn, norm_eps, norm_num_groups,
                                          temporal_compression_ratio=temporal_compression_ratio)
        self.quant_conv = None
        self.post_quant_conv = None
        self.use_slicing = False
        self.use_tiling = False
        self.num_latent_frames_batch_size = 2          # reference :1079
        down = 2 ** (len(block_out_channels) - 1)      # tile geometry, reference :1081-1098
        self.tile_sample_min_height = sample_height // 2
        self.tile_sample_min_width = sample_width // 2
        self.tile_latent_min_height = int(self.tile_sample_min_height / down)
        self.tile_latent_min_width = int(self.tile_sample_min_width / down)
        self.tile_overlap_factor_height = 1 / 6
        self.tile_overlap_factor_width = 1 / 5

    # ---- tiling / slicing knobs (:1109-1174) ----
    def enable_tiling(self, tile_sample_min_height: Optional[int] = None, tile_sample_min_width: Optional[int] = None,
                      tile_overlap_factor_height: Optional[float] = None, tile_overlap_factor_width: Optional[float] = None) -> None:
        """reference :1109-1153 (`x or current`).  288 GB of HBM never needs it; it changes the RESULT (every tile sees zero
        padding at its own border and the seams are ramps), so it is built for callers that ran the reference with it on."""
        down = 2 ** (len(self.config.block_out_channels) - 1)
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        self.tile_latent_min_height = int(self.tile_sample_min_height / down)
        self.tile_latent_min_width = int(self.tile_sample_min_width / down)
        self.tile_overlap_factor_height = tile_overlap_factor_height or self.tile_overlap_factor_height
        self.tile_overlap_factor_width = tile_overlap_factor_width or self.tile_overlap_factor_width

    def disable_tiling(self):
        self.use_tiling = False

    def enable_slicing(self):
        """reference :1162-1167: decode one batch element at a time (:1274-1278).  Every kernel of this decoder treats batch
        elements independently, so the sliced and the batched decode are the same bits; the flag is honoured as a no-op."""
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def _clear_fake_context_parallel_cache(self):
        for m in self.modules():
            if isinstance(m, CogVideoXCausalConv3d):
                m._clear_fake_context_parallel_cache()

    # ---- decode (:1217-1280) ----
    def _decode_cl(self, z: torch.Tensor, frames_out: Optional[torch.Tensor] = None, scale: float = 1.0):
        """z [N,16,T,h,w] bf16 -> list of channels-last chunks (or fills `frames_out` fp32 [N,3,T',H,W])."""
        if not z.is_cuda or z.dtype != BF16 or self.dtype != BF16:
            raise TcxError(f"AutoencoderKLCogVideoX.decode: needs bf16 latents and weights on the GPU "
                           f"(got {z.dtype} on {z.device}, weights {self.dtype}); no CPU fallback")
        N, C, T, h, w = z.shape
        zcl = ops.ncthw_to_cl(z, scale)          # layout change + the 1/scaling_factor of decode_latents (:512), one rounding
        fbs = self.num_latent_frames_batch_size
        if self.use_tiling and (w > self.tile_latent_min_width or h > self.tile_latent_min_height):      # :1222-1225
            assert frames_out is None                    # decode_to_frames sizes its output from the tiled result
            return [self._tiled_decode_cl(zcl)]
        if T == 1:
            bounds = [(0, 1)]
        else:
            rem = T % fbs
            bounds = [(fbs * i + (0 if i == 0 else rem), fbs * (i + 1) + rem) for i in range(T // fbs)]
        self._clear_fake_context_parallel_cache()
        chunks, t_off = [], 0
        for s, e in bounds:
            d = self.decoder.forward_cl(zcl[:, s:e].contiguous())
            if frames_out is not None:
                ops.cl_to_frames(d, frames_out, t_off)
                t_off += d.shape[1]
            else:
                chunks.append(d)
        self._clear_fake_context_parallel_cache()
        return chunks

    def _tiled_decode_cl(self, zcl: torch.Tensor) -> torch.Tensor:
        """reference `tiled_decode` :1303-1392 on the channels-last latent [N,T,h,w,16]: every spatial tile goes through the
        decoder chunk by chunk with its own conv cache, is blended IN PLACE with the already blended tile above and to its left
        (`tcx_blend_ramp_bf16`), cropped to the row limits and concatenated -> [N,T',H',W',3] bf16."""
        N, T, h, w, _ = zcl.shape
        fbs = self.num_latent_frames_batch_size
        if T // fbs == 0:
            raise ValueError(f"tiled decode of {T} latent frame(s): the reference's tile loop (:1345-1364) runs over "
                             f"num_frames // {fbs} = 0 chunks and fails on an empty concat; call disable_tiling() for single frames")
        lh, lw = self.tile_latent_min_height, self.tile_latent_min_width
        overlap_h = int(lh * (1 - self.tile_overlap_factor_height))
        overlap_w = int(lw * (1 - self.tile_overlap_factor_width))
        if overlap_h <= 0 or overlap_w <= 0:
            raise ValueError(f"tiled decode: tile stride {overlap_h} x {overlap_w} latent rows / columns (the reference's `range` "
                             f"step, :1341-1343) must be positive")
        blend_h = int(self.tile_sample_min_height * self.tile_overlap_factor_height)
        blend_w = int(self.tile_sample_min_width * self.tile_overlap_factor_width)
        limit_h = self.tile_sample_min_height - blend_h
        limit_w = self.tile_sample_min_width - blend_w
        rem = T % fbs
        bounds = [(fbs * k + (0 if k == 0 else rem), fbs * (k + 1) + rem) for k in range(T // fbs)]
        rows = []
        for i in range(0, h, overlap_h):
            row = []
            for j in range(0, w, overlap_w):
                self._clear_fake_context_parallel_cache()
                time = [self.decoder.forward_cl(zcl[:, s:e, i:i + lh, j:j + lw].contiguous()) for s, e in bounds]
                row.append(torch.cat(time, dim=1))
            rows.append(row)
        self._clear_fake_context_parallel_cache()
        result_rows = []
        for i, row in enumerate(rows):
            result_row = []
            for j, tile in enumerate(row):
                if i > 0:
                    ops.blend_ramp(rows[i - 1][j], tile, blend_h, 2)
                if j > 0:
                    ops.blend_ramp(row[j - 1], tile, blend_w, 3)
                result_row.append(tile[:, :, :limit_h, :limit_w])
            result_rows.append(torch.cat(result_row, dim=3))
        return torch.cat(result_rows, dim=2)

    # the reference's names for the pieces of decode (:1217-1253, :1282-1392), on its NCTHW tensors
    @torch.no_grad()
    def _decode(self, z: torch.Tensor, return_dict: bool = True):
        return self.decode(z, return_dict=return_dict)

    @torch.no_grad()
    def tiled_decode(self, z: torch.Tensor, return_dict: bool = True):
        """reference :1303-1392 regardless of `use_tiling` / the size test of `_decode` (:1222-1225)."""
        if not z.is_cuda or z.dtype != BF16 or self.dtype != BF16:
            raise TcxError("AutoencoderKLCogVideoX.tiled_decode: needs bf16 latents and weights on the GPU; no CPU fallback")
        dec = self._tiled_decode_cl(ops.ncthw_to_cl(z, 1.0)).permute(0, 4, 1, 2, 3).contiguous()
        return DecoderOutput(sample=dec) if return_dict else (dec,)

    def blend_v(self, a: torch.Tensor, b: torch.Tensor, blend_extent: int) -> torch.Tensor:
        """reference :1282-1291 on [N,C,T,H,W] bf16 tensors: the first rows of `b` ramp from the last rows of `a` (in place on b)."""
        return self._blend(a, b, blend_extent, 2)

    def blend_h(self, a: torch.Tensor, b: torch.Tensor, blend_extent: int) -> torch.Tensor:
        """reference :1293-1301: the same along the width."""
        return self._blend(a, b, blend_extent, 3)

    def _blend(self, a, b, blend_extent, dim_cl):
        acl, bcl = a.permute(0, 2, 3, 4, 1).contiguous(), b.permute(0, 2, 3, 4, 1).contiguous()
        ops.blend_ramp(acl, bcl, blend_extent, dim_cl)
        b.copy_(bcl.permute(0, 4, 1, 2, 3))
        return b

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        chunks = self._decode_cl(z)
        dec = torch.cat(chunks, dim=1).permute(0, 4, 1, 2, 3).contiguous()          # -> [N,3,T',H,W]
        if not return_dict:
            return (dec,)
        return DecoderOutput(sample=dec)

    def decoded_frames(self, T: int) -> int:
        """Frames `decode` returns for T latent frames: the chunks of `_decode` (:1235-1241: remainder folded into the first) each go
        through log2(temporal_compression_ratio) temporal upsamplings of diffusers CogVideoXUpsample3D(compress_time): 1 frame stays 1,
        an odd count t > 1 becomes 2 t - 1 (first frame kept), an even count 2 t.  Odd T (the pipeline's 4 k + 1 frame clips):
        4 (T - 1) + 1; even T: 4 T."""
        fbs, levels = self.num_latent_frames_batch_size, int(np.log2(self.config.temporal_compression_ratio))
        chunks = [1] if T == 1 else [fbs + (T % fbs if i == 0 else 0) for i in range(T // fbs)]
        total = 0
        for t in chunks:
            for _ in range(levels):
                t = t if t == 1 else (2 * t - 1 if t % 2 else 2 * t)
            total += t
        return total

    @torch.no_grad()
    def decode_to_frames(self, z: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """decode + `(x/2+.5).clamp(0,1).float()` of pipeline decode_latents (:514-517), written chunk by chunk."""
        N, C, T, h, w = z.shape
        if self.use_tiling and (w > self.tile_latent_min_width or h > self.tile_latent_min_height):
            return self.cl_to_frames(self.decode_cl_bf16(z, scale))
        sf = 2 ** (len(self.config.block_out_channels) - 1)
        Tout = self.decoded_frames(T)
        frames = torch.empty((N, self.config.out_channels, Tout, h * sf, w * sf), device=z.device, dtype=torch.float32)
        self._decode_cl(z, frames, scale)
        return frames

    @torch.no_grad()
    def decode_cl_bf16(self, z: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """decode -> the decoder's own bf16 output, channels-last [N,T',H,W,3]: what the data-parallel runner all-gathers
        (half the bytes of the fp32 frames and lossless: `cl_to_frames` of it is bit-identical to `decode_to_frames`)."""
        return torch.cat(self._decode_cl(z, None, scale), dim=1)

    @torch.no_grad()
    def cl_to_frames(self, x_cl: torch.Tensor) -> torch.Tensor:
        """channels-last bf16 decoder output [N,T,H,W,C] -> fp32 frames [N,C,T,H,W] = (x/2+.5).clamp(0,1) (:514-517)."""
        N, T, H, W, C = x_cl.shape
        frames = torch.empty((N, C, T, H, W), device=x_cl.device, dtype=torch.float32)
        ops.cl_to_frames(x_cl.contiguous(), frames, 0)
        return frames

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_dict: bool = True,
                generator: Optional[torch.Generator] = None):
        """reference :1394-1410: the autoencoding round trip encode -> posterior sample / mode -> decode.  Like the reference it returns
        what `decode` returns (a DecoderOutput), or a 1-tuple of it with return_dict=False."""
        posterior = self.encode(sample).latent_dist
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        dec = self.decode(z)
        if not return_dict:
            return (dec,)
        return dec

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """reference :1176-1215: x [N,3,F,H,W] bf16 in [-1,1] -> posterior over [N,16,T,H/8,W/8] (4-frame chunks
        with the remainder folded into the first, conv caches carried across chunks)."""
        if not x.is_cuda or x.dtype != BF16 or self.dtype != BF16:
            raise TcxError(f"AutoencoderKLCogVideoX.encode: needs bf16 pixels and weights on the GPU "
                           f"(got {x.dtype} on {x.device}, weights {self.dtype}); no CPU fallback")
        N, C, Fr, H, W = x.shape
        if H % 8 or W % 8:
            raise ValueError(f"encode: height and width must be divisible by 8, got {H}x{W}")
        xcl = _pad_channels(ops.ncthw_to_cl(x))
        self._clear_fake_context_parallel_cache()
        if Fr == 1:
            bounds = [(0, 1)]
        else:
            fbs, rem = 4, Fr % 4
            bounds = [(fbs * i + (0 if i == 0 else rem), fbs * (i + 1) + rem) for i in range(Fr // fbs)]
        hs = [self.encoder.forward_cl(xcl[:, s:e].contiguous()) for s, e in bounds]
        self._clear_fake_context_parallel_cache()
        moments = torch.cat(hs, dim=1).permute(0, 4, 1, 2, 3).contiguous()          # -> [N,32,T,h,w]
        posterior = DiagonalGaussianDistribution(moments)
        if not return_dict:
            return (posterior,)
        return AutoencoderKLOutput(latent_dist=posterior)
