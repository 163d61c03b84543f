"""Thin torch-tensor front end over the C ABI (libtcx_hip.so).

torch is used for device memory and streams; every hot-path op, the GEMMs included (`gemm_bf16`), is a
hand-written HIP kernel reached through ctypes.  Nothing here falls back to torch arithmetic: a missing
library or a failing launch raises `TcxError`.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import TCX_BF16, TCX_F32, TcxError, check

BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, name: str, dtype=BF16) -> None:
    if not t.is_cuda:
        raise TcxError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TcxError(f"{name}: expected dtype {dtype}, got {t.dtype}")


def _bshd_strides(t: torch.Tensor, name: str) -> Tuple[int, int, int]:
    """[B, S, H, D] view -> (stride_b, stride_s, stride_h); last dim must be contiguous."""
    if t.dim() != 4 or t.stride(3) != 1:
        raise TcxError(f"{name}: expected a [B,S,H,D] view with contiguous D, got {tuple(t.shape)} / {t.stride()}")
    return t.stride(0), t.stride(1), t.stride(2)


# ----------------------------------------------------------------------------- attention
_ATTN_TIMING = None      # None (off) or {head_dim: [(start_event, stop_event), ...]}


def attn_timing_start() -> None:
    """Bracket every tcx_attn_fwd launch with HIP events on its launch stream (bench.py roofline leg)."""
    global _ATTN_TIMING
    _ATTN_TIMING = {}


def attn_timing_stop() -> dict:
    """-> {head_dim: {"n": launches, "ms": total device milliseconds}}; synchronises."""
    global _ATTN_TIMING
    rec, _ATTN_TIMING = _ATTN_TIMING or {}, None
    torch.cuda.synchronize()
    return {d: {"n": len(ev), "ms": float(sum(a.elapsed_time(b) for a, b in ev))} for d, ev in rec.items()}


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float,
             out: Optional[torch.Tensor] = None, out_dtype=BF16, log2_scores: bool = False,
             k_sqmax: Optional[torch.Tensor] = None, split_tail: bool = True, bound_proven: bool = False) -> torch.Tensor:
    """q [B,Sq,H,D], k/v [B,Sk,H,D] bf16 views -> o [B,Sq,H,D].

    log2_scores: q k^T already is the base-2 exponent (q pre-multiplied by scale*log2(e)); scale must be 1.
    k_sqmax: fp32 [B,H] max_k |k|^2 from `qk_layernorm_rope` (log2_scores + D=64 only): bound-centred softmax loop.
    split_tail: let the bound-centred D = 64 launch split its last partly filled round of workgroups along the keys
    (needs a scratch buffer, allocated here; False = the single-pass launch, for A/B runs and tests).
    bound_proven: the caller guarantees |q_row| * sqrt(k_sqmax) * 1.002 + 1e-3 < 60 for every row (TCX_ATTN_BOUND_PROVEN): no
    per-workgroup test, no launch of the exact kernel on the complement."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _need(t, n)
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    if k.shape != (B, Sk, H, D) or v.shape != (B, Sk, H, D):
        raise TcxError(f"attn_fwd: shape mismatch q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    if out is None:
        out = torch.empty((B, Sq, H, D), device=q.device, dtype=out_dtype)
    _need(out, "out", out_dtype)
    if k_sqmax is not None:
        _need(k_sqmax, "k_sqmax", torch.float32)
        if tuple(k_sqmax.shape) != (B, H) or not k_sqmax.is_contiguous():
            raise TcxError(f"attn_fwd: k_sqmax must be contiguous fp32 [{B},{H}], got {tuple(k_sqmax.shape)}")
    lib = _lib.load()
    ev = None
    if _ATTN_TIMING is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    flags = (_lib.TCX_ATTN_LOG2_SCORES if log2_scores else 0) | (_lib.TCX_ATTN_BOUND_PROVEN if bound_proven else 0)
    odt = TCX_F32 if out_dtype == torch.float32 else TCX_BF16
    # scratch of the tail split (balances the last, partly filled round of workgroups; include/tcx_hip.h): caller-owned
    ws_bytes = int(lib.tcx_attn_fwd_workspace_bytes(B, H, Sq, Sk, D, flags, int(k_sqmax is not None), odt)) if split_tail else 0
    ws = torch.empty((ws_bytes,), device=q.device, dtype=torch.uint8) if ws_bytes > 0 else None
    check(lib.tcx_attn_fwd_ws(_p(q), _p(k), _p(v), _p(out), B, H, Sq, Sk, D,
                              *_bshd_strides(q, "q"), *_bshd_strides(k, "k"), *_bshd_strides(v, "v"),
                              *_bshd_strides(out, "out"), float(scale), flags, _p(k_sqmax), odt, _p(ws), ws_bytes, _stream()),
          "tcx_attn_fwd")
    if ev is not None:
        ev[1].record()
        _ATTN_TIMING.setdefault(D, []).append(ev)
    return out


def qk_layernorm_rope(q, k, gq, bq, gk, bk, cos, sin, text_len: int, eps: float = 1e-6, q_scale: float = 1.0,
                      want_k_sqmax: bool = False) -> Optional[torch.Tensor]:
    """In-place per-head LayerNorm + RoPE on q, k [B,S,H,64] bf16 views; q additionally times q_scale.
    want_k_sqmax: also return fp32 [B,H] = max over tokens of |k|^2 (input of attn_fwd's bound-centred loop)."""
    _need(q, "q"); _need(k, "k")
    B, S, H, D = q.shape
    sq = _bshd_strides(q, "q")
    if _bshd_strides(k, "k") != sq or k.shape != q.shape:
        raise TcxError("qk_layernorm_rope: q and k must share shape and strides")
    if cos is not None:
        _need(cos, "cos", torch.float32); _need(sin, "sin", torch.float32)
        if cos.shape != (S - text_len, D) or not cos.is_contiguous() or not sin.is_contiguous():
            raise TcxError(f"qk_layernorm_rope: cos/sin must be contiguous [{S - text_len},{D}], got {tuple(cos.shape)}")
    lib = _lib.load()
    ksq = torch.empty((B, H), device=q.device, dtype=torch.float32) if want_k_sqmax else None
    check(lib.tcx_qk_layernorm_rope(_p(q), _p(k), B, S, H, D, *sq, _p(gq), _p(bq), _p(gk), _p(bk), _p(cos), _p(sin),
                                    text_len, float(eps), float(q_scale), _p(ksq), _stream()), "tcx_qk_layernorm_rope")
    return ksq


def _rows3(t: torch.Tensor, name: str) -> Tuple[int, int, int, int]:
    """[B, rows, C] view with contiguous rows -> (B, rows, C, stride_b)."""
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != t.shape[2]:
        raise TcxError(f"{name}: expected [B,rows,C] with contiguous rows, got {tuple(t.shape)} / {t.stride()}")
    return t.shape[0], t.shape[1], t.shape[2], t.stride(0)


def layernorm_modulate(x, gamma, beta, eps: float, shift_v=None, scale_v=None, shift_t=None, scale_t=None,
                       text_len: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = LN(x)*gamma+beta [* (1+scale) + shift]; modulation vectors are [B, C] views (row stride free)."""
    _need(x, "x")
    B, rows, Cc, xsb = _rows3(x, "x")
    if out is None:
        out = torch.empty((B, rows, Cc), device=x.device, dtype=BF16)
    _, _, _, ysb = _rows3(out, "out")
    msb = 0
    for m in (shift_v, scale_v, shift_t, scale_t):
        if m is not None:
            _need(m, "modulation")
            if m.shape != (B, Cc) or m.stride(1) != 1:
                raise TcxError(f"layernorm_modulate: modulation must be [B,C] with contiguous C, got {tuple(m.shape)}")
            if msb not in (0, m.stride(0)):
                raise TcxError("layernorm_modulate: modulation vectors must share one batch stride")
            msb = m.stride(0)
    lib = _lib.load()
    check(lib.tcx_layernorm_modulate(_p(x), _p(out), B, rows, Cc, xsb, ysb, _p(gamma), _p(beta), _p(shift_v), _p(scale_v),
                                     _p(shift_t), _p(scale_t), msb, text_len, float(eps), _stream()), "tcx_layernorm_modulate")
    return out


def gated_residual_(x, y, gate_v=None, gate_t=None, text_len: int = 0) -> torch.Tensor:
    """x += gate * y in place (gate [B,C] views, text rows first)."""
    _need(x, "x"); _need(y, "y")
    B, rows, Cc, xsb = _rows3(x, "x")
    By, rowsy, Cy, ysb = _rows3(y, "y")
    if (By, rowsy, Cy) != (B, rows, Cc):
        raise TcxError(f"gated_residual_: shape mismatch {tuple(x.shape)} vs {tuple(y.shape)}")
    gsb = 0
    for g in (gate_v, gate_t):
        if g is not None:
            _need(g, "gate")
            if g.shape != (B, Cc) or g.stride(1) != 1:
                raise TcxError("gated_residual_: gate must be [B,C] with contiguous C")
            gsb = g.stride(0)
    lib = _lib.load()
    check(lib.tcx_gated_residual(_p(x), _p(y), B, rows, Cc, xsb, ysb, _p(gate_v), _p(gate_t), gsb, text_len, _stream()),
          "tcx_gated_residual")
    return x


def bias_gelu_tanh_(x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need(x, "x")
    if not x.is_contiguous():
        raise TcxError("bias_gelu_tanh_: x must be contiguous")
    Cc = x.shape[-1]
    lib = _lib.load()
    check(lib.tcx_bias_gelu_tanh(_p(x), _p(bias), _p(x), x.numel() // Cc, Cc, _stream()), "tcx_bias_gelu_tanh")
    return x


def scale_bf16(x: torch.Tensor, s: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need(x, "x")
    if not x.is_contiguous():
        raise TcxError("scale_bf16: x must be contiguous")
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().tcx_scale_bf16(_p(x), _p(out), x.numel(), float(s), _stream()), "tcx_scale_bf16")
    return out


def scale_sqmax(x: torch.Tensor, s: float, H: int, D: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """x: [B, S, H*D] bf16 view (unit inner stride, free row / batch strides) -> (y = bf16(x * s) contiguous [B,S,H*D],
    sqmax fp32 [B,H] = max over rows of |y[b,s,h,:]|^2): k * scale of the cross-attention + attn_fwd's k_sqmax in one pass."""
    _need(x, "x")
    if x.dim() != 3 or x.shape[2] != H * D or x.stride(2) != 1:
        raise TcxError(f"scale_sqmax: expected a [B,S,{H * D}] view with unit inner stride, got {tuple(x.shape)} / {x.stride()}")
    B, S, _ = x.shape
    y = torch.empty((B, S, H * D), device=x.device, dtype=BF16)
    sq = torch.empty((B, H), device=x.device, dtype=torch.float32)
    check(_lib.load().tcx_scale_sqmax_bf16(_p(x), _p(y), B, S, H, D, x.stride(0), x.stride(1), float(s), _p(sq), _stream()),
          "tcx_scale_sqmax_bf16")
    return y, sq


def silu(x: torch.Tensor) -> torch.Tensor:
    _need(x, "x")
    x = x.contiguous()
    out = torch.empty_like(x)
    check(_lib.load().tcx_silu_bf16(_p(x), _p(out), x.numel(), _stream()), "tcx_silu_bf16")
    return out


def patchify(a: torch.Tensor, b: Optional[torch.Tensor], p: int, k_pad: int = 1) -> torch.Tensor:
    """[B,F,Ca,H,W] (+ [B,F,Cb,H,W]) -> [B*F*(H/p)*(W/p), K], K = (Ca+Cb)*p*p rounded up to a multiple of k_pad
    (zero-filled: `gemm_bf16` wants K % 128 == 0)."""
    _need(a, "a")
    a = a.contiguous()
    B, F, Ca, H, W = a.shape
    Cb = 0
    if b is not None:
        _need(b, "b")
        b = b.contiguous()
        if b.shape[:2] != (B, F) or b.shape[3:] != (H, W):
            raise TcxError(f"patchify: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        Cb = b.shape[2]
    K = (Ca + Cb) * p * p
    ks = (K + k_pad - 1) // k_pad * k_pad
    out = torch.empty((B * F * (H // p) * (W // p), ks), device=a.device, dtype=BF16)
    check(_lib.load().tcx_patchify(_p(a), _p(b), _p(out), B, F, Ca, Cb, H, W, p, ks, _stream()), "tcx_patchify")
    return out


def unpatchify(x: torch.Tensor, B: int, F: int, Cout: int, H: int, W: int, p: int, out_dtype=BF16) -> torch.Tensor:
    _need(x, "x")
    x = x.contiguous()
    out = torch.empty((B, F, Cout, H, W), device=x.device, dtype=out_dtype)
    check(_lib.load().tcx_unpatchify(_p(x), _p(out), B, F, Cout, H, W, p,
                                     TCX_F32 if out_dtype == torch.float32 else TCX_BF16, _stream()), "tcx_unpatchify")
    return out


def cfg_ddim_step(uncond: torch.Tensor, cond: Optional[torch.Tensor], x: torch.Tensor, guidance: float,
                  alpha_t: float, alpha_prev: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need(x, "x")
    if uncond.dtype not in (BF16, torch.float32):
        raise TcxError(f"cfg_ddim_step: prediction dtype {uncond.dtype} unsupported")
    if not (uncond.is_contiguous() and x.is_contiguous() and (cond is None or cond.is_contiguous())):
        raise TcxError("cfg_ddim_step: tensors must be contiguous")
    if uncond.numel() != x.numel() or (cond is not None and (cond.numel() != x.numel() or cond.dtype != uncond.dtype)):
        raise TcxError("cfg_ddim_step: size / dtype mismatch")
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().tcx_cfg_ddim_step(_p(uncond), _p(cond), _p(x), _p(out), x.numel(), float(guidance), float(alpha_t),
                                        float(alpha_prev), TCX_F32 if uncond.dtype == torch.float32 else TCX_BF16, _stream()),
          "tcx_cfg_ddim_step")
    return out


def cfg_ddim_eta_step(uncond: torch.Tensor, cond: Optional[torch.Tensor], x: torch.Tensor, guidance: float, coef, noise: torch.Tensor) -> torch.Tensor:
    """CFG + `DDIMScheduler.step(eta > 0)`, fused (`tcx_cfg_ddim_eta_step`); coef = (sqrt a_t, sqrt(1 - a_t), sqrt a_prev, direction
    coefficient, std_dev) from the scheduler, noise = its fp32 variance noise."""
    _need(x, "x"); _need(noise, "noise", torch.float32)
    if uncond.dtype not in (BF16, torch.float32):
        raise TcxError(f"cfg_ddim_eta_step: prediction dtype {uncond.dtype} unsupported")
    if not (uncond.is_contiguous() and x.is_contiguous() and noise.is_contiguous() and (cond is None or cond.is_contiguous())):
        raise TcxError("cfg_ddim_eta_step: tensors must be contiguous")
    if uncond.numel() != x.numel() or noise.numel() != x.numel() or (cond is not None and (cond.numel() != x.numel() or cond.dtype != uncond.dtype)):
        raise TcxError("cfg_ddim_eta_step: size / dtype mismatch")
    out = torch.empty_like(x)
    check(_lib.load().tcx_cfg_ddim_eta_step(_p(uncond), _p(cond), _p(x), _p(out), x.numel(), float(guidance), *[float(v) for v in coef],
                                            _p(noise), TCX_F32 if uncond.dtype == torch.float32 else TCX_BF16, _stream()), "tcx_cfg_ddim_eta_step")
    return out


def cfg_ddim_cog_step(uncond: torch.Tensor, cond: Optional[torch.Tensor], x: torch.Tensor, guidance: float, sqrt_alpha_t: float,
                      sqrt_beta_t: float, coef_sample: float, coef_x0: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CFG + `CogVideoXDDIMScheduler.step` (sampler "DDIM_Cog"), fused; coefficients from the scheduler's float64 tables."""
    _need(x, "x")
    if uncond.dtype not in (BF16, torch.float32):
        raise TcxError(f"cfg_ddim_cog_step: prediction dtype {uncond.dtype} unsupported")
    if not (uncond.is_contiguous() and x.is_contiguous() and (cond is None or cond.is_contiguous())):
        raise TcxError("cfg_ddim_cog_step: tensors must be contiguous")
    if uncond.numel() != x.numel() or (cond is not None and (cond.numel() != x.numel() or cond.dtype != uncond.dtype)):
        raise TcxError("cfg_ddim_cog_step: size / dtype mismatch")
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().tcx_cfg_ddim_cog_step(_p(uncond), _p(cond), _p(x), _p(out), x.numel(), float(guidance), float(sqrt_alpha_t),
                                            float(sqrt_beta_t), float(coef_sample), float(coef_x0),
                                            TCX_F32 if uncond.dtype == torch.float32 else TCX_BF16, _stream()), "tcx_cfg_ddim_cog_step")
    return out


def cfg_sigma_step(kind: int, uncond: torch.Tensor, cond: Optional[torch.Tensor], x: torch.Tensor, guidance: float, coef,
                   hist_in: Optional[torch.Tensor] = None, hist_out: Optional[torch.Tensor] = None,
                   noise: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CFG + the `step` of a sigma-parametrised sampler, fused (`tcx_cfg_sigma_step`): kind `_lib.TCX_STEP_EULER` ("Euler", and
    "Euler A" with `noise`) or `_lib.TCX_STEP_DPMPP_2M` ("DPM++": `hist_out` receives this step's x0, `hist_in` is the previous
    one).  `coef`: the 5 fp32 scalars of include/tcx_hip.h, from the scheduler."""
    import ctypes
    _need(x, "x")
    if uncond.dtype not in (BF16, torch.float32):
        raise TcxError(f"cfg_sigma_step: prediction dtype {uncond.dtype} unsupported")
    if not (uncond.is_contiguous() and x.is_contiguous() and (cond is None or cond.is_contiguous())):
        raise TcxError("cfg_sigma_step: tensors must be contiguous")
    if uncond.numel() != x.numel() or (cond is not None and (cond.numel() != x.numel() or cond.dtype != uncond.dtype)):
        raise TcxError("cfg_sigma_step: size / dtype mismatch")
    for name, t in (("hist_in", hist_in), ("hist_out", hist_out), ("noise", noise)):
        if t is not None:
            _need(t, name, torch.float32)
            if not t.is_contiguous() or t.numel() != x.numel():
                raise TcxError(f"cfg_sigma_step: {name} must be a contiguous fp32 tensor of the latents' size")
    if len(coef) != 5:
        raise TcxError("cfg_sigma_step: coef holds 5 scalars")
    carr = (ctypes.c_float * 5)(*[float(v) for v in coef])
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().tcx_cfg_sigma_step(_p(uncond), _p(cond), _p(x), _p(out), x.numel(), float(guidance), int(kind),
                                         ctypes.cast(carr, ctypes.c_void_p), _p(hist_in), _p(hist_out), _p(noise),
                                         TCX_F32 if uncond.dtype == torch.float32 else TCX_BF16, _stream()), "tcx_cfg_sigma_step")
    return out


def cfg_pndm_step(mode: int, uncond: torch.Tensor, cond: Optional[torch.Tensor], x: torch.Tensor, guidance: float, coef,
                  e1: Optional[torch.Tensor] = None, e2: Optional[torch.Tensor] = None, e3: Optional[torch.Tensor] = None,
                  cur_in: Optional[torch.Tensor] = None, cur_out: Optional[torch.Tensor] = None,
                  mo_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CFG + one `PNDMScheduler.step` (Runge-Kutta evaluation or 4th-order multistep update), fused (`tcx_cfg_pndm_step`); `coef`:
    the 6 fp32 scalars of include/tcx_hip.h; history / accumulator tensors fp32 of the latents' size."""
    import ctypes
    _need(x, "x")
    if uncond.dtype not in (BF16, torch.float32):
        raise TcxError(f"cfg_pndm_step: prediction dtype {uncond.dtype} unsupported")
    if not (uncond.is_contiguous() and x.is_contiguous() and (cond is None or cond.is_contiguous())):
        raise TcxError("cfg_pndm_step: tensors must be contiguous")
    if uncond.numel() != x.numel() or (cond is not None and (cond.numel() != x.numel() or cond.dtype != uncond.dtype)):
        raise TcxError("cfg_pndm_step: size / dtype mismatch")
    for name, t in (("e1", e1), ("e2", e2), ("e3", e3), ("cur_in", cur_in), ("cur_out", cur_out), ("mo_out", mo_out)):
        if t is not None:
            _need(t, name, torch.float32)
            if not t.is_contiguous() or t.numel() != x.numel():
                raise TcxError(f"cfg_pndm_step: {name} must be a contiguous fp32 tensor of the latents' size")
    if len(coef) != 6:
        raise TcxError("cfg_pndm_step: coef holds 6 scalars")
    carr = (ctypes.c_float * 6)(*[float(v) for v in coef])
    out = torch.empty_like(x)
    check(_lib.load().tcx_cfg_pndm_step(_p(uncond), _p(cond), _p(x), _p(out), x.numel(), float(guidance), int(mode),
                                        ctypes.cast(carr, ctypes.c_void_p), _p(e1), _p(e2), _p(e3), _p(cur_in), _p(cur_out), _p(mo_out),
                                        TCX_F32 if uncond.dtype == torch.float32 else TCX_BF16, _stream()), "tcx_cfg_pndm_step")
    return out


def div_bf16(x: torch.Tensor, d: float) -> torch.Tensor:
    """bf16(x / d) (`tcx_div_bf16`): the Euler samplers' `scale_model_input`."""
    _need(x, "x")
    if not x.is_contiguous():
        raise TcxError("div_bf16: x must be contiguous")
    y = torch.empty_like(x)
    check(_lib.load().tcx_div_bf16(_p(x), _p(y), x.numel(), float(d), _stream()), "tcx_div_bf16")
    return y


# ----------------------------------------------------------------------------- VAE (channels-last [N,T,H,W,C])
def conv3d_cl(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], cache: Optional[torch.Tensor] = None,
              res: Optional[torch.Tensor] = None, ups: int = 0, t_map: Optional[torch.Tensor] = None,
              stride: int = 1, pad: Optional[Tuple[int, int]] = None, out_hw: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """x [N,T,H,W,Cin], w [Cout,kT,kH,kW,Cin] (pre-permuted), cache [N,kT-1,H,W,Cin] -> y [N,T',H',W',Cout].

    Default: "same" convolution (pad = k // 2 on both sides).  `stride=2, pad=(0, 0), out_hw=(H//2, W//2)` is the
    encoder's downsample conv (zero row / column appended at the bottom / right by the range check)."""
    _need(x, "x"); _need(w, "w")
    if not (x.is_contiguous() and w.is_contiguous()):
        raise TcxError("conv3d_cl: x and w must be contiguous")
    N, T, H, W, Cin = x.shape
    Cout, kT, kH, kW, Cin2 = w.shape
    if Cin2 != Cin:
        raise TcxError(f"conv3d_cl: Cin mismatch {Cin} vs {Cin2}")
    T_out = T if t_map is None else t_map.numel()
    if cache is not None:
        _need(cache, "cache")
        if tuple(cache.shape) != (N, kT - 1, H, W, Cin) or not cache.is_contiguous():
            raise TcxError(f"conv3d_cl: cache must be contiguous [N,{kT - 1},H,W,Cin], got {tuple(cache.shape)}")
    if t_map is not None and (t_map.dtype != torch.int32 or not t_map.is_cuda):
        raise TcxError("conv3d_cl: t_map must be a GPU int32 tensor")
    ph, pw = (kH // 2, kW // 2) if pad is None else pad
    Ho, Wo = ((H << ups), (W << ups)) if out_hw is None else out_hw
    y = torch.empty((N, T_out, Ho, Wo, Cout), device=x.device, dtype=BF16)
    if res is not None:
        _need(res, "res")
        if res.shape != y.shape or not res.is_contiguous():
            raise TcxError(f"conv3d_cl: residual must be contiguous {tuple(y.shape)}, got {tuple(res.shape)}")
    check(_lib.load().tcx_conv3d_cl(_p(x), _p(cache), _p(w), _p(bias), _p(res), _p(y), N, T, H, W, Cin, Cout, kT, kH, kW,
                                    T_out, ups, stride, ph, pw, Ho, Wo, _p(t_map), _stream()), "tcx_conv3d_cl")
    return y


def conv3d_route(Cin: int, Cout: int, k: Tuple[int, int, int] = (3, 3, 3), ups: int = 0, stride: int = 1,
                 T: int = 2, H: int = 64, W: int = 64, t_map: bool = False, res: bool = False) -> int:
    """Kernel `conv3d_cl` launches for this shape (TCX_CONV_ROUTE_*: 1 mfma 256x256, 2 mfma 512x128, 3 narrow, 4 igemm);
    host-only query of the dispatch function itself."""
    Ho, Wo = (H << ups, W << ups) if stride == 1 else ((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1)
    return int(_lib.load().tcx_conv3d_route(T, H, W, Cin, Cout, k[0], k[1], k[2], ups, stride, Ho, Wo, int(t_map or ups == 1), int(res)))


def avgpool_t(x: torch.Tensor) -> torch.Tensor:
    """Temporal average pool of CogVideoXDownsample3D(compress_time): x [N,T,H,W,C] -> [N,T',H,W,C]."""
    _need(x, "x")
    if not x.is_contiguous():
        raise TcxError("avgpool_t: x must be contiguous")
    N, T, H, W, Cc = x.shape
    To = T // 2 if T % 2 == 0 else 1 + (T - 1) // 2
    y = torch.empty((N, To, H, W, Cc), device=x.device, dtype=BF16)
    check(_lib.load().tcx_avgpool_t(_p(x), _p(y), N, T, H * W, Cc, _stream()), "tcx_avgpool_t")
    return y


def blend_ramp(a: torch.Tensor, b: torch.Tensor, blend_extent: int, dim: int) -> torch.Tensor:
    """`blend_v` (dim 2) / `blend_h` (dim 3) of the tiled VAE decode (reference autoencoder_magvit.py:1282-1301) on channels-last
    tiles [N,T,H,W,C] bf16: the first rows / columns of `b` become a ramp between the last ones of `a` and themselves, in place."""
    _need(a, "a"); _need(b, "b")
    if dim not in (2, 3) or a.dim() != 5 or b.dim() != 5 or not a.is_contiguous() or not b.is_contiguous():
        raise TcxError("blend_ramp: contiguous channels-last tiles [N,T,H,W,C] and dim 2 (rows) or 3 (columns)")
    other = 3 if dim == 2 else 2
    if a.shape[:2] != b.shape[:2] or a.shape[4] != b.shape[4] or a.shape[other] != b.shape[other]:
        raise TcxError(f"blend_ramp: tiles {tuple(a.shape)} and {tuple(b.shape)} do not share the seam")
    ext = min(a.shape[dim], b.shape[dim], int(blend_extent))
    if ext <= 0:
        return b
    N, T, Hb, Wb, C = b.shape
    Ha, Wa = a.shape[2], a.shape[3]
    if dim == 2:
        outer, inner = N * T, Wb * C
        a_so, a_se, b_so, b_se = Ha * Wa * C, Wa * C, Hb * Wb * C, Wb * C
        a_off = (Ha - ext) * Wa * C
    else:
        outer, inner = N * T * Hb, C
        a_so, a_se, b_so, b_se = Wa * C, C, Wb * C, C
        a_off = (Wa - ext) * C
    check(_lib.load().tcx_blend_ramp_bf16(a.data_ptr() + 2 * a_off, _p(b), outer, ext, inner, a_so, a_se, b_so, b_se, _stream()),
          "tcx_blend_ramp_bf16")
    return b


GEMM_BIAS, GEMM_BIAS_GELU, GEMM_GATED_RESIDUAL = 0, 1, 2


def gemm_supported(N: int, K: int) -> bool:
    """Shapes `tcx_gemm_bf16` takes (16-byte rows of W / X and 16-byte stores); K % 128 != 0 runs its K-tail variant."""
    return N % 8 == 0 and K % 8 == 0


def _gemm_rows(t: torch.Tensor, name: str, N: int, M: int) -> Tuple[int, int, int]:
    """A [.., N] tensor of M rows -> (rows_per_batch, ld, stride_b).  rows_per_batch = 0: all rows share one stride
    (flat); otherwise a [B, rows, N] view whose batch stride is free (a row range of the joint text+video buffer)."""
    if t.dim() < 1 or t.shape[-1] != N or t.stride(-1) != 1 or t.numel() != M * N:
        raise TcxError(f"gemm_bf16: {name} must be [.., {N}] with {M} rows and a contiguous last dimension, got {tuple(t.shape)} / {t.stride()}")
    dims = [(n_, st) for n_, st in zip(t.shape[:-1], t.stride()[:-1]) if n_ != 1]
    if not dims:
        return 0, N, 0
    ld = dims[-1][1]
    if all(dims[i][1] == dims[i + 1][1] * dims[i + 1][0] for i in range(len(dims) - 1)):
        return 0, ld, 0
    if len(dims) == 2:
        return dims[1][0], ld, dims[0][1]
    raise TcxError(f"gemm_bf16: {name} rows are neither uniformly strided nor a [B, rows, N] view: {tuple(t.shape)} / {t.stride()}")


def gemm_bf16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = GEMM_BIAS,
              res: Optional[torch.Tensor] = None, gate_v: Optional[torch.Tensor] = None,
              gate_t: Optional[torch.Tensor] = None, text_len: int = 0,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = epilogue(x @ w.T + bias) on the hand-written 256x256 MFMA kernel (include/tcx_hip.h, tcx_gemm_bf16).
    x [..., K] bf16 (rows uniformly strided, K contiguous), w [N, K] bf16 contiguous.
    GEMM_GATED_RESIDUAL: y = res + gate * (acc + bias); res / out [.., N] or [B, rows, N] views with a free batch stride
    (a row range of the joint buffer); gate_v / gate_t [B, N] views (unit inner stride, common batch stride) or both
    None; rows r < text_len of every batch item take gate_t.  `out` may be `res` (in place)."""
    _need(x, "x"); _need(w, "w")
    if w.dim() != 2 or not w.is_contiguous() or x.shape[-1] != w.shape[1]:
        raise TcxError(f"gemm_bf16: x {tuple(x.shape)} does not match w {tuple(w.shape)} (contiguous [N,K])")
    N, K = w.shape
    M = x.numel() // K
    xr, ldx, _ = _gemm_rows(x, "x", K, M)
    if xr:
        raise TcxError("gemm_bf16: the rows of x must be uniformly strided")
    if out is None:
        out = torch.empty((*x.shape[:-1], N), device=x.device, dtype=torch.bfloat16)
    _need(out, "out")
    rpb, ldy, ysb = _gemm_rows(out, "out", N, M)
    if bias is not None:
        _need(bias, "bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise TcxError("gemm_bf16: bias must be a contiguous [N] tensor")
    ldres, rsb, gsb = 0, 0, 0
    if epilogue == GEMM_GATED_RESIDUAL:
        if res is None:
            raise TcxError("gemm_bf16: the gated-residual epilogue needs res")
        _need(res, "res")
        rpb_r, ldres, rsb = _gemm_rows(res, "res", N, M)
        if (gate_v is None) != (gate_t is None):
            raise TcxError("gemm_bf16: give both gates or none")
        rpb_g = 0
        if gate_v is not None:
            for n_, g in (("gate_v", gate_v), ("gate_t", gate_t)):
                _need(g, n_)
                if g.dim() != 2 or g.shape[1] != N or g.stride(1) != 1:
                    raise TcxError(f"gemm_bf16: {n_} must be a [B, N] view with unit inner stride")
            B = gate_v.shape[0]
            if gate_v.stride(0) != gate_t.stride(0) or gate_t.shape[0] != B or M % B != 0:
                raise TcxError("gemm_bf16: gate batch geometry does not match x")
            rpb_g, gsb = M // B, gate_v.stride(0)
        want = {r for r in (rpb, rpb_r, rpb_g) if r}
        if len(want) > 1:
            raise TcxError(f"gemm_bf16: out / res / gates disagree on rows per batch: {sorted(want)}")
        rpb = want.pop() if want else 0
        if rpb and not rsb:
            rsb = rpb * ldres                                            # flat res seen as [B, rpb, N]
    if rpb and not ysb:
        ysb = rpb * ldy
    check(_lib.load().tcx_gemm_bf16(_p(x), _p(w), _p(bias), _p(out), M, N, K, ldx, ldy, ysb, int(epilogue), _p(res), ldres, rsb,
                                    _p(gate_v), _p(gate_t), gsb, int(rpb), int(text_len), _stream()), "tcx_gemm_bf16")
    return out


def groupnorm_stats(x: torch.Tensor, groups: int, eps: float) -> torch.Tensor:
    """x [N,T,H,W,C] channels-last -> stats fp32 [N,G,2] (mean, rstd)."""
    _need(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    S = x.numel() // (N * Cc)
    nsplit = int(max(1, min(2048, S // 64)))
    stats = torch.empty((N, groups, 2), device=x.device, dtype=torch.float32)
    partial = torch.empty((N, nsplit, 2, Cc), device=x.device, dtype=torch.float32)
    check(_lib.load().tcx_groupnorm_stats(_p(x), _p(stats), _p(partial), N, S, Cc, groups, float(eps), nsplit, _stream()),
          "tcx_groupnorm_stats")
    return stats


def groupnorm_apply(x, stats, gn_w, gn_b, groups: int, ytab=None, btab=None, z_t_map=None, silu: bool = True,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need(x, "x")
    N, T, H, W, Cc = x.shape
    Tz = Hz = Wz = 0
    if ytab is not None:
        _need(ytab, "ytab"); _need(btab, "btab")
        if ytab.shape != btab.shape or ytab.shape[0] != N or ytab.shape[-1] != Cc:
            raise TcxError("groupnorm_apply: bad modulation tables")
        Tz, Hz, Wz = ytab.shape[1:4]
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().tcx_groupnorm_spatialnorm_silu(_p(x), _p(out), _p(stats), _p(gn_w), _p(gn_b), _p(ytab), _p(btab),
                                                     N, T, H, W, Cc, groups, Tz, Hz, Wz, _p(z_t_map), int(silu), _stream()),
          "tcx_groupnorm_spatialnorm_silu")
    return out


def ncthw_to_cl(x: torch.Tensor, mul: float = 1.0) -> torch.Tensor:
    """[N,C,T,H,W] bf16 -> channels-last [N,T,H,W,C] bf16 (scaled by mul)."""
    _need(x, "x")
    x = x.contiguous()
    N, Cc, T, H, W = x.shape
    y = torch.empty((N, T, H, W, Cc), device=x.device, dtype=BF16)
    check(_lib.load().tcx_ncthw_to_cl(_p(x), _p(y), N, Cc, T * H * W, float(mul), _stream()), "tcx_ncthw_to_cl")
    return y


def cl_to_frames(x: torch.Tensor, out: torch.Tensor, t_offset: int) -> None:
    """channels-last [N,T,H,W,C] bf16 -> out[:, :, t_offset:t_offset+T] fp32 = clamp(x/2+.5, 0, 1); out [N,C,Ttot,H,W]."""
    _need(x, "x"); _need(out, "out", torch.float32)
    N, T, H, W, Cc = x.shape
    if out.shape[0] != N or out.shape[1] != Cc or out.shape[3:] != (H, W) or not out.is_contiguous():
        raise TcxError("cl_to_frames: bad output tensor")
    check(_lib.load().tcx_cl_to_ncthw_frames(_p(x), _p(out), N, Cc, T * H * W, out.shape[2] * H * W, t_offset * H * W,
                                             _stream()), "tcx_cl_to_ncthw_frames")


# ----------------------------------------------------------------------------- point-cloud render
def bilinear_splat(src: torch.Tensor, mask1: Optional[torch.Tensor], depth: torch.Tensor, flow: torch.Tensor, is_image: bool,
                   flow_scale: float = 1.0):
    """`Warper.bilinear_splatting` (reference models/utils.py:422-583): src [b,c<=4,h,w], mask1 [b,1,h,w] | None, depth [b,h,w],
    flow [b,2,h,w] (times flow_scale) -> (out [b,c,h,w], mask2 [b,1,h,w]); all fp32, GPU, contiguous."""
    for n, t in (("src", src), ("depth", depth), ("flow", flow)) + ((("mask1", mask1),) if mask1 is not None else ()):
        _need(t, n, torch.float32)
        if not t.is_contiguous():
            raise TcxError(f"bilinear_splat: {n} must be contiguous")
    b, c, h, w = src.shape
    if not 1 <= c <= 4 or tuple(depth.shape) != (b, h, w) or tuple(flow.shape) != (b, 2, h, w) or (mask1 is not None and tuple(mask1.shape) != (b, 1, h, w)):
        raise TcxError("bilinear_splat: shape mismatch (src [b,c<=4,h,w], depth [b,h,w], flow [b,2,h,w], mask1 [b,1,h,w])")
    f32 = dict(device=src.device, dtype=torch.float32)
    acc = torch.empty((b * (h + 2) * (w + 2) * 5 + 1,), **f32)
    out, mask2 = torch.empty((b, c, h, w), **f32), torch.empty((b, 1, h, w), **f32)
    check(_lib.load().tcx_bilinear_splat(_p(src), _p(mask1), _p(depth), _p(flow), _p(acc), _p(out), _p(mask2), b, c, h, w,
                                         1 if is_image else 0, float(flow_scale), _stream()), "tcx_bilinear_splat")
    return out, mask2


def warp_forward(frame: torch.Tensor, mask1: Optional[torch.Tensor], depth: torch.Tensor, mats: torch.Tensor,
                 per_item_max: bool = False, clean_points: bool = False, return_tdepth: bool = False):
    """frame [b,3,h,w], depth [b,1,h,w], mask1 [b,1,h,w] | None, mats [b,30] (all fp32, GPU, contiguous)
    -> (warped [b,3,h,w], mask2 [b,1,h,w], warped_depth [b,1,h,w], flow [b,2,h,w]).
    per_item_max: each batch item is rendered as its own batch-1 reference call (TCX_WARP_PER_ITEM_MAX);
    clean_points: forward_warp(mask=True) (TCX_WARP_CLEAN_POINTS)."""
    for n, t in (("frame", frame), ("depth", depth), ("mats", mats)) + ((("mask1", mask1),) if mask1 is not None else ()):
        _need(t, n, torch.float32)
        if not t.is_contiguous():
            raise TcxError(f"warp_forward: {n} must be contiguous")
    b, c, h, w = frame.shape
    if c != 3 or tuple(depth.shape) != (b, 1, h, w) or tuple(mats.shape) != (b, 30) or (mask1 is not None and tuple(mask1.shape) != (b, 1, h, w)):
        raise TcxError("warp_forward: shape mismatch")
    f32 = dict(device=frame.device, dtype=torch.float32)
    flow, tdepth = torch.empty((b, 2, h, w), **f32), torch.empty((b, h, w), **f32)
    acc = torch.empty((b * (h + 2) * (w + 2) * 5 + b,), **f32)
    warped, mask2, wdepth = torch.empty((b, 3, h, w), **f32), torch.empty((b, 1, h, w), **f32), torch.empty((b, 1, h, w), **f32)
    check(_lib.load().tcx_warp_forward(_p(frame), _p(mask1), _p(depth), _p(mats), _p(flow), _p(tdepth), _p(acc), _p(warped),
                                       _p(mask2), _p(wdepth), b, h, w, (1 if per_item_max else 0) | (2 if clean_points else 0), _stream()), "tcx_warp_forward")
    if return_tdepth:
        return warped, mask2, wdepth, flow, tdepth
    return warped, mask2, wdepth, flow
