"""Precision policy of the oracle (test infrastructure, see oracle/__init__.py).

All arithmetic is done in float32 on the CPU; the policy only decides where a
value is rounded to the activation dtype.

``R`` – contract rounding point: a tensor the HIP path materialises in HBM.
``r`` – per-op rounding point: a tensor the *reference's eager bf16 execution*
        materialises but the fused HIP kernels keep in registers.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class Prec:
    MODES = ("fp32", "bf16", "bf16_ref")

    def __init__(self, mode: str = "fp32"):
        if mode not in self.MODES:
            raise ValueError(f"unknown precision mode {mode!r}; expected one of {self.MODES}")
        self.mode = mode
        self.act_dtype = torch.float32 if mode == "fp32" else torch.bfloat16

    # -- rounding points -------------------------------------------------
    def R(self, x: torch.Tensor) -> torch.Tensor:
        """Contract rounding point (HIP path writes this tensor to HBM)."""
        if self.mode == "fp32":
            return x.float()
        return x.to(torch.bfloat16).float()

    def r(self, x: torch.Tensor) -> torch.Tensor:
        """Per-op rounding point of the reference's eager bf16 execution."""
        if self.mode == "bf16_ref":
            return x.to(torch.bfloat16).float()
        return x.float()

    def param(self, w: torch.Tensor | None) -> torch.Tensor | None:
        """Parameters are stored in the activation dtype (model.to(bf16))."""
        if w is None:
            return None
        return w.to(self.act_dtype).float()

    # -- basic ops (fp32 arithmetic, one rounding at the output) ----------
    def linear(self, x, w, b=None):
        return self.R(F.linear(x.float(), self.param(w), self.param(b)))

    def linear_fused(self, x, w, b=None):
        """A Linear whose result the HIP path keeps in registers for a fused epilogue (`res + gate * y` in
        tcx_gemm_bf16): rounded only where the reference's eager execution materialises it."""
        return self.r(F.linear(x.float(), self.param(w), self.param(b)))

    def layer_norm(self, x, w, b, eps, contract_point: bool = False):
        y = F.layer_norm(x.float(), (x.shape[-1],), self.param(w), self.param(b), eps)
        return self.R(y) if contract_point else self.r(y)

    def out(self, x: torch.Tensor) -> torch.Tensor:
        """Cast a float32 working tensor to the externally visible dtype."""
        return x.to(self.act_dtype)
