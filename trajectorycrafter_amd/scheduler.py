"""Host-side DDIM scheduler with the diffusers `DDIMScheduler` surface the reference pipeline uses
(`set_timesteps`, `timesteps`, `step`, `scale_model_input`, `init_noise_sigma`, `order`, `config`).

Selected by the reference at demo.py:647-657 (`DDIM_Origin`), used at
models/pipeline_trajectorycrafter.py:846,1099,1164-1167.  The checkpoint's scheduler_config.json is
not available offline; the defaults below are the CogVideoX-Fun-V1.1-5b-InP values as recalled
(unverified) and every one of them is overridable / loadable from a config dict.

Coefficient tables live on the host (fp32, numpy/torch CPU); the per-step tensor update is the fused
HIP kernel `tcx_cfg_ddim_step` (CFG combine + v-prediction DDIM update + bf16 rounding).
"""
from __future__ import annotations

import json
import os
from typing import Optional

import numpy as np
import torch

from .config import FrozenConfig


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "v_prediction",
                 timestep_spacing: str = "trailing", rescale_betas_zero_snr: bool = True, set_alpha_to_one: bool = True,
                 steps_offset: int = 0, clip_sample: bool = False, **unused):
        if beta_schedule != "scaled_linear":
            raise ValueError(f"beta_schedule {beta_schedule!r} not supported")
        if prediction_type != "v_prediction":
            raise ValueError("the fused HIP step implements v_prediction (CogVideoX); got " + prediction_type)
        if clip_sample:
            raise ValueError("clip_sample=True is not supported")
        self.config = FrozenConfig(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                   beta_schedule=beta_schedule, prediction_type=prediction_type,
                                   timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr,
                                   set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, clip_sample=clip_sample)
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
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "scheduler_config.json")) as f:
            return cls(**{k: v for k, v in json.load(f).items() if not k.startswith("_")})

    def set_timesteps(self, num_inference_steps: int, device=None):
        N = self.config.num_train_timesteps
        if num_inference_steps > N:
            raise ValueError(f"num_inference_steps ({num_inference_steps}) > num_train_timesteps ({N})")
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing == "trailing":
            ts = np.round(np.arange(N, 0, -N / num_inference_steps)).astype(np.int64) - 1
        elif self.config.timestep_spacing == "leading":
            ts = (np.arange(0, num_inference_steps) * (N // num_inference_steps)).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        else:
            raise ValueError(f"timestep_spacing {self.config.timestep_spacing!r} not supported")
        self.timesteps = torch.from_numpy(ts)            # host ints; the pipeline never syncs on them

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coeffs(self, timestep: int):
        prev = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[timestep])
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps) -> torch.Tensor:
        """diffusers `DDIMScheduler.add_noise` (used by the reference's `strength < 1` branch, pipeline :431-436): the schedule is
        cast to the SAMPLE dtype first and every operation runs in it — sqrt(abar_t) x0 + sqrt(1 - abar_t) noise with bf16
        intermediates for bf16 latents.  Torch elementwise ops on the latent tensor, once per clip (conditioning preparation)."""
        abar = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        t = torch.as_tensor(timesteps, device=original_samples.device).reshape(-1).long()
        sa = abar[t] ** 0.5
        sb = (1 - abar[t]) ** 0.5
        while sa.dim() < original_samples.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise

    accepts_eta = True           # the library's `step` has an `eta` parameter (the pipeline passes its own `eta` only to such schedulers, :521-540)

    def eta_coeffs(self, timestep: int, eta: float):
        """(sqrt a_t, sqrt(1 - a_t), sqrt a_prev, sqrt(1 - a_prev - std^2), std) for `step(eta > 0)`: the library's expressions on 0-dim
        fp32 tensors — variance = (1 - a_prev)/(1 - a_t) (1 - a_t/a_prev), std = eta sqrt(variance)."""
        prev = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep].float()
        a_prev = (self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod).float()
        b_t, b_prev = 1 - a_t, 1 - a_prev
        variance = (b_prev / b_t) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        # eta = 1 at the zero-SNR first step (a_t = 0): 1 - a_prev - std^2 is exactly 0 in real arithmetic and +-1 ulp in fp32 (the schedule
        # table itself differs in the last bit between hosts) — clamped at 0, where the library would take the root of a negative number
        return [float(a_t ** 0.5), float(b_t ** 0.5), float(a_prev ** 0.5), float(torch.clamp(1 - a_prev - std ** 2, min=0) ** 0.5), float(std)]

    def fused_cfg_step(self, uncond: torch.Tensor, cond: Optional[torch.Tensor], sample: torch.Tensor, guidance: float,
                       timestep: int, generator=None, eta: float = 0.0) -> torch.Tensor:
        """CFG combine (pipeline :1157-1161) + `step` + the bf16 cast (:1178) as ONE kernel (`tcx_cfg_ddim_step`; eta > 0:
        `tcx_cfg_ddim_eta_step` with the variance noise drawn like the library's randn_tensor, fp32, from `generator`)."""
        from . import ops
        if eta == 0.0:
            a_t, a_prev = self.coeffs(int(timestep))
            return ops.cfg_ddim_step(uncond, cond, sample, guidance, a_t, a_prev)
        gdev = generator.device if generator is not None else sample.device
        noise = torch.randn(sample.shape, generator=generator, device=gdev, dtype=torch.float32).to(sample.device)
        return ops.cfg_ddim_eta_step(uncond, cond, sample, guidance, self.eta_coeffs(int(timestep), float(eta)), noise)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0,
             generator=None, return_dict: bool = False):
        """diffusers-shaped step (no guidance): prev_sample = DDIM(model_output, sample), bf16 on the GPU."""
        return (self.fused_cfg_step(model_output.contiguous(), None, sample.contiguous(), 1.0, int(timestep), generator=generator, eta=eta),)


class CogVideoXDDIMScheduler(DDIMScheduler):
    """diffusers `CogVideoXDDIMScheduler` — the reference's "DDIM_Cog" sampler (demo.py:652).  Same timestep grid and the
    same DDIM (eta = 0) update as `DDIMScheduler`, with three differences, all kept: the noise schedule is held in float64
    and SNR-shifted (`alphas_cumprod / (s + (1 - s) alphas_cumprod)`, s = `snr_shift_scale`) BEFORE the zero-terminal-SNR rescale,
    which is applied to alphas_cumprod directly; and `step` is written in the `a_t x + b_t x0` form, whose bf16 rounding points
    differ from the `sqrt(a_prev) x0 + sqrt(1 - a_prev) eps` form (`tcx_cfg_ddim_cog_step`).  Defaults: the class defaults of
    diffusers except the values CogVideoX-5b's scheduler_config.json overrides (as recalled, SURVEY §8c: v_prediction,
    trailing, zero-SNR, snr_shift_scale 1.0 for the 5B family — 3.0, the class default, is the 2B value)."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "v_prediction",
                 timestep_spacing: str = "trailing", rescale_betas_zero_snr: bool = True, set_alpha_to_one: bool = True,
                 steps_offset: int = 0, clip_sample: bool = False, snr_shift_scale: float = 1.0, **unused):
        super().__init__(num_train_timesteps, beta_start, beta_end, beta_schedule, prediction_type, timestep_spacing,
                         rescale_betas_zero_snr, set_alpha_to_one, steps_offset, clip_sample)
        self.config = FrozenConfig(dict(self.config), snr_shift_scale=snr_shift_scale)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        abar = torch.cumprod(1.0 - betas, dim=0)
        abar = abar / (snr_shift_scale + (1 - snr_shift_scale) * abar)                    # SNR shift (SD3-style)
        if rescale_betas_zero_snr:
            r = abar.sqrt()
            r0, rT = r[0].clone(), r[-1].clone()
            abar = ((r - rT) * (r0 / (r0 - rT))) ** 2
        self.betas = betas
        self.alphas_cumprod = abar                                                       # float64
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64) if set_alpha_to_one else abar[0]

    def step_coeffs(self, timestep: int):
        """(sqrt(a_t), sqrt(1 - a_t), coef_sample, coef_x0) in float64 arithmetic, as python floats."""
        a_t, a_prev = self.coeffs(int(timestep))
        ca = ((1 - a_prev) / (1 - a_t)) ** 0.5
        return a_t ** 0.5, (1 - a_t) ** 0.5, ca, a_prev ** 0.5 - a_t ** 0.5 * ca

    def fused_cfg_step(self, uncond, cond, sample, guidance: float, timestep: int, generator=None, eta: float = 0.0) -> torch.Tensor:
        from . import ops
        # The library's `CogVideoXDDIMScheduler.step` HAS an `eta` parameter (so the reference pipeline forwards its `eta` to it,
        # :521-540) and never reads it: the update is the deterministic `a x + b x0` form whatever eta is.  Same here: accepted
        # and unused (round 3 refused a non-zero value; the reference would have run).
        del eta
        sa, sb, ca, cb = self.step_coeffs(timestep)
        return ops.cfg_ddim_cog_step(uncond, cond, sample, guidance, sa, sb, ca, cb)


# ---------------------------------------------------------------------------------------------------------------------------
# The sigma-parametrised samplers of the reference's table (demo.py:647-654): "Euler", "Euler A", "DPM++".
# diffusers EulerDiscreteScheduler / EulerAncestralDiscreteScheduler / DPMSolverMultistepScheduler as `from_pretrained` builds
# them from the CogVideoX scheduler_config.json (scaled_linear betas, v_prediction, trailing spacing, zero-terminal-SNR rescale —
# these classes then pin alphas_cumprod[-1] to 2^-24, i.e. sigma_max = 4096; unknown keys such as snr_shift_scale are ignored by the
# library and here).  Restated from the published algorithms (diffusers is absent offline: parity unpinned, like the DDIM classes).
# Host side: fp32 tables and, per step, the library's 0-dim fp32 coefficient expressions evaluated with torch in the library's
# order; device side: ONE fused kernel per step (`tcx_cfg_sigma_step`: guidance + step + bf16 cast) and, for the Euler pair,
# `tcx_div_bf16` for `scale_model_input`.
# ---------------------------------------------------------------------------------------------------------------------------
class _SigmaSampler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "v_prediction", timestep_spacing: str = "trailing",
                 rescale_betas_zero_snr: bool = True, steps_offset: int = 0, **unused):
        if beta_schedule != "scaled_linear":
            raise ValueError(f"beta_schedule {beta_schedule!r} not supported")
        if prediction_type != "v_prediction":
            raise ValueError("the fused HIP step implements v_prediction (CogVideoX); got " + prediction_type)
        if timestep_spacing not in ("trailing", "leading"):
            raise ValueError(f"timestep_spacing {timestep_spacing!r} not supported")
        self.config = FrozenConfig(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                   beta_schedule=beta_schedule, prediction_type=prediction_type, timestep_spacing=timestep_spacing,
                                   rescale_betas_zero_snr=rescale_betas_zero_snr, steps_offset=steps_offset)
        base = DDIMScheduler(num_train_timesteps, beta_start, beta_end, beta_schedule, prediction_type, "trailing",
                             rescale_betas_zero_snr)                       # the same fp32 beta / zero-SNR arithmetic
        self.betas = base.betas
        self.alphas_cumprod = base.alphas_cumprod.clone()
        if rescale_betas_zero_snr:
            self.alphas_cumprod[-1] = 2 ** -24                              # finite first sigma (4096) instead of inf
        self._train_sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        self.sigmas: Optional[torch.Tensor] = None
        self.timesteps: Optional[torch.Tensor] = None
        self.num_inference_steps: Optional[int] = None

    from_pretrained = classmethod(DDIMScheduler.from_pretrained.__func__)

    def _spaced(self, n: int) -> np.ndarray:
        N = self.config.num_train_timesteps
        if n > N:
            raise ValueError(f"num_inference_steps ({n}) > num_train_timesteps ({N})")
        if self.config.timestep_spacing == "trailing":
            return np.round(np.arange(N, 0, -N / n)) - 1
        return (np.arange(0, n) * (N // n)).round()[::-1].copy().astype(np.float64) + self.config.steps_offset

    def _set_tables(self, n: int, ts_dtype) -> None:
        ts = self._spaced(n).astype(ts_dtype)
        sig = np.interp(ts, np.arange(0, len(self._train_sigmas)), self._train_sigmas)       # interpolation_type "linear"
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))     # final_sigmas_type "zero"
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n
        self._index = {int(t): i for i, t in enumerate(ts.tolist())}
        if len(self._index) != len(ts):
            raise ValueError("duplicate timesteps in the schedule (more inference steps than distinct train steps) are not supported")

    def _i(self, timestep) -> int:
        if self.sigmas is None:
            raise RuntimeError("call set_timesteps first")
        try:
            return self._index[int(timestep)]
        except KeyError:
            raise ValueError(f"timestep {timestep} is not on the schedule set by set_timesteps({self.num_inference_steps})") from None

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps) -> torch.Tensor:
        """The library's `add_noise` (the `strength < 1` start, pipeline :431-436): the sigma table cast to the SAMPLE dtype, every
        operation in it — Euler pair: x0 + noise sigma; DPM++: alpha x0 + sig noise (alpha = 1 / sqrt(sigma^2 + 1), sig = sigma alpha,
        evaluated in that dtype as well).  Torch elementwise ops on the latent tensor, once per clip."""
        t = torch.as_tensor(timesteps).reshape(-1).tolist()
        idx = torch.tensor([self._i(v) for v in t], device=original_samples.device)
        sigma = self.sigmas.to(device=original_samples.device, dtype=original_samples.dtype)[idx].flatten()
        while sigma.dim() < original_samples.dim():
            sigma = sigma.unsqueeze(-1)
        if isinstance(self, DPMSolverMultistepScheduler):
            alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
            sigma_t = sigma * alpha_t
            return alpha_t * original_samples + sigma_t * noise
        return original_samples + noise * sigma

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None, return_dict: bool = False, **kw):
        """diffusers-shaped step (no guidance) on the fused kernel."""
        return (self.fused_cfg_step(model_output.contiguous(), None, sample.contiguous(), 1.0, timestep, generator=generator),)


class EulerDiscreteScheduler(_SigmaSampler):
    """ "Euler": x_in = x / sqrt(sigma^2 + 1);  x0 = v (-sigma / sqrt(sigma^2 + 1)) + x / (sigma^2 + 1);  d = (x - x0) / sigma;
    x_next = x + d (sigma_next - sigma), fp32, then the loop's bf16 cast.  s_churn = 0 (the pipeline passes none)."""
    ancestral = False

    def set_timesteps(self, num_inference_steps: int, device=None):
        self._set_tables(num_inference_steps, np.float32)                   # timestep_type "discrete": float32 timesteps

    @property
    def init_noise_sigma(self) -> float:
        mx = float(self.sigmas.max()) if self.sigmas is not None else float(self._train_sigmas.max())
        return mx if self.config.timestep_spacing == "trailing" else float((torch.tensor(mx) ** 2 + 1) ** 0.5)

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        from . import ops
        sigma = self.sigmas[self._i(timestep)]
        return ops.div_bf16(sample.contiguous(), float((sigma ** 2 + 1) ** 0.5))

    def step_coeffs(self, timestep):
        i = self._i(timestep)
        sigma, sigma_to = self.sigmas[i], self.sigmas[i + 1]
        a, s2p1 = -sigma / (sigma ** 2 + 1) ** 0.5, sigma ** 2 + 1
        if not self.ancestral:
            return [float(a), float(s2p1), float(sigma), float(sigma_to - sigma), 0.0]
        sigma_up = (sigma_to ** 2 * (sigma ** 2 - sigma_to ** 2) / sigma ** 2) ** 0.5
        sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
        return [float(a), float(s2p1), float(sigma), float(sigma_down - sigma), float(sigma_up)]

    def fused_cfg_step(self, uncond, cond, sample, guidance: float, timestep, generator=None) -> torch.Tensor:
        from . import _lib, ops
        noise = None
        if self.ancestral:      # randn_tensor(model_output.shape, dtype=model_output.dtype (fp32 in the loop), generator=generator)
            gdev = generator.device if generator is not None else sample.device
            noise = torch.randn(sample.shape, generator=generator, device=gdev, dtype=torch.float32).to(sample.device)
        return ops.cfg_sigma_step(_lib.TCX_STEP_EULER, uncond, cond, sample, guidance, self.step_coeffs(timestep), noise=noise)


class EulerAncestralDiscreteScheduler(EulerDiscreteScheduler):
    """ "Euler A": the Euler step to sigma_down, then + N(0, 1) sigma_up (one fp32 draw of the latents' shape per step from the
    call's generator), with sigma_up^2 = sigma_next^2 (sigma^2 - sigma_next^2) / sigma^2 and sigma_down^2 = sigma_next^2 - sigma_up^2."""
    ancestral = True


class DPMSolverMultistepScheduler(_SigmaSampler):
    """ "DPM++": DPM-Solver++ 2M (midpoint), first-order on the first step and on the last (final sigma 0); keeps the previous
    step's data prediction x0 (fp32, the latents' size) between steps.  init_noise_sigma = 1, int64 timesteps."""
    init_noise_sigma = 1.0
    solver_order = 2

    def set_timesteps(self, num_inference_steps: int, device=None):
        self._set_tables(num_inference_steps, np.int64)
        self._hist = [None, None]                                           # [previous x0, scratch for this step's]
        self._lower_order_nums = 0

    @staticmethod
    def _alpha_sig(sigma: torch.Tensor):
        alpha = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha, sigma * alpha

    def step_coeffs(self, timestep):
        """([alpha_i, sig_i, A, B, 1/r0], second_order) — the library's expressions on 0-dim fp32 tensors."""
        i, n = self._i(timestep), len(self.timesteps)
        alpha_t, sig_t = self._alpha_sig(self.sigmas[i + 1])
        alpha_s, sig_s = self._alpha_sig(self.sigmas[i])
        lam_t, lam_s = torch.log(alpha_t) - torch.log(sig_t), torch.log(alpha_s) - torch.log(sig_s)
        h = lam_t - lam_s
        A, B = sig_t / sig_s, alpha_t * (torch.exp(-h) - 1.0)
        if self._lower_order_nums < 1 or i == n - 1:                        # warm-up step / lower_order_final (final sigma 0)
            return [float(alpha_s), float(sig_s), float(A), float(B), 0.0], False
        alpha_p, sig_p = self._alpha_sig(self.sigmas[i - 1])
        r0 = (lam_s - (torch.log(alpha_p) - torch.log(sig_p))) / h
        return [float(alpha_s), float(sig_s), float(A), float(B), float(1.0 / r0)], True

    def fused_cfg_step(self, uncond, cond, sample, guidance: float, timestep, generator=None) -> torch.Tensor:
        from . import _lib, ops
        coef, second = self.step_coeffs(timestep)
        prev_x0, scratch = self._hist
        if second and (prev_x0 is None or prev_x0.shape != sample.shape or prev_x0.device != sample.device):
            raise RuntimeError("DPM++: the previous step's data prediction is missing (steps must run in schedule order after set_timesteps)")
        if scratch is None or scratch.shape != sample.shape or scratch.device != sample.device:
            scratch = torch.empty(sample.shape, device=sample.device, dtype=torch.float32)
        out = ops.cfg_sigma_step(_lib.TCX_STEP_DPMPP_2M, uncond, cond, sample, guidance, coef,
                                 hist_in=prev_x0 if second else None, hist_out=scratch)
        self._hist = [scratch, prev_x0]                                     # ping-pong: this step's x0 becomes "previous"
        if self._lower_order_nums < self.solver_order:
            self._lower_order_nums += 1
        return out


class PNDMScheduler:
    """ "PNDM" (demo.py:651): diffusers `PNDMScheduler` as `from_pretrained` builds it from the CogVideoX scheduler config — scaled-linear
    betas WITHOUT the zero-terminal-SNR rescale (the class has no such option and ignores the key), v_prediction, trailing spacing,
    set_alpha_to_one, skip_prk_steps = False: `set_timesteps(50)` yields 59 timesteps (12 Runge-Kutta evaluations over the first three
    intervals, then 47 fourth-order multistep updates) and the pipeline loops over all of them.  Restated from the published algorithm
    (parity unpinned).  Host: the fp32 alpha table, the evaluation counter and which history tensors exist; device: one fused kernel
    per step (`tcx_cfg_pndm_step`) + four fp32 history tensors and one accumulator of the latents' size."""
    order = 1
    init_noise_sigma = 1.0
    pndm_order = 4

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "v_prediction", timestep_spacing: str = "trailing",
                 set_alpha_to_one: bool = True, steps_offset: int = 0, skip_prk_steps: bool = False, **unused):
        if beta_schedule != "scaled_linear":
            raise ValueError(f"beta_schedule {beta_schedule!r} not supported")
        if prediction_type != "v_prediction":
            raise ValueError("the fused HIP step implements v_prediction (CogVideoX); got " + prediction_type)
        if skip_prk_steps:
            raise ValueError("skip_prk_steps=True is not built (the library's default, and what the reference's table gets, is False)")
        if timestep_spacing not in ("trailing", "leading"):
            raise ValueError(f"timestep_spacing {timestep_spacing!r} not supported")
        self.config = FrozenConfig(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                   beta_schedule=beta_schedule, prediction_type=prediction_type, timestep_spacing=timestep_spacing,
                                   set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, skip_prk_steps=skip_prk_steps)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.timesteps: Optional[torch.Tensor] = None
        self.num_inference_steps: Optional[int] = None

    from_pretrained = classmethod(DDIMScheduler.from_pretrained.__func__)

    def set_timesteps(self, num_inference_steps: int, device=None):
        N, n = self.config.num_train_timesteps, num_inference_steps
        if n > N:
            raise ValueError(f"num_inference_steps ({n}) > num_train_timesteps ({N})")
        if n < self.pndm_order:
            raise ValueError(f"PNDM with Runge-Kutta warm-up needs at least {self.pndm_order} inference steps, got {n}")
        self.num_inference_steps = n
        if self.config.timestep_spacing == "trailing":
            grid = np.round(np.arange(N, 0, -N / n))[::-1].astype(np.int64) - 1
        else:
            grid = (np.arange(0, n) * (N // n)).round().astype(np.int64) + self.config.steps_offset
        half = N // n // 2
        prk = np.array(grid[-self.pndm_order:]).repeat(2) + np.tile(np.array([0, half]), self.pndm_order)
        self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
        self.plms_timesteps = grid[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self._counter = 0
        self._ets: list = []                 # newest last, at most 4 fp32 tensors (rotated, never reallocated after the fourth)
        self._cur = None                     # Runge-Kutta running sum (fp32)
        self._cur_sample = None              # the sample a Runge-Kutta group started from

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps) -> torch.Tensor:
        """The library's `add_noise`: sqrt(a_t) x0 + sqrt(1 - a_t) noise with the table cast to the sample dtype (as DDIMScheduler's)."""
        return DDIMScheduler.add_noise(self, original_samples, noise, timesteps)

    def prev_coeffs(self, timestep: int, prev_timestep: int):
        """[sqrt(a_t), sqrt(1 - a_t), sqrt(a_prev / a_t), a_prev - a_t, a_t sqrt(1 - a_prev) + sqrt(a_t (1 - a_t) a_prev)]: the library's
        expressions of `_get_prev_sample` on 0-dim fp32 tensors."""
        a = self.alphas_cumprod[timestep]
        ap = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b, bp = 1 - a, 1 - ap
        return [float(a ** 0.5), float(b ** 0.5), float((ap / a) ** 0.5), float(ap - a), float(a * bp ** 0.5 + (a * b * ap) ** 0.5)]

    def _buf(self, like: torch.Tensor) -> torch.Tensor:
        return torch.empty(like.shape, device=like.device, dtype=torch.float32)

    def fused_cfg_step(self, uncond, cond, sample, guidance: float, timestep, generator=None) -> torch.Tensor:
        from . import _lib, ops
        if self.timesteps is None:
            raise RuntimeError("call set_timesteps first")
        N, n = self.config.num_train_timesteps, self.num_inference_steps
        t = int(timestep)
        if self._counter >= len(self.timesteps):
            raise RuntimeError("PNDM: more steps than set_timesteps scheduled (the schedule is stateful: call set_timesteps again)")
        if t != int(self.timesteps[self._counter]):
            raise ValueError(f"PNDM: step {self._counter} expects timestep {int(self.timesteps[self._counter])}, got {t} "
                             "(the schedule is stateful: steps must follow `timesteps` in order)")
        if self._counter < len(self.prk_timesteps):                      # a Runge-Kutta evaluation (library: step_prk)
            k = self._counter % 4
            prev_t = t - (0 if self._counter % 2 else N // n // 2)
            group_t = int(self.prk_timesteps[self._counter // 4 * 4])
            coef = self.prev_coeffs(group_t, prev_t)
            if k == 0:
                self._cur_sample = sample
                self._cur = self._buf(sample)
                mo = self._buf(sample)
                out = ops.cfg_pndm_step(_lib.TCX_PNDM_PRK_FIRST, uncond, cond, self._cur_sample, guidance, [1 / 6] + coef,
                                        cur_out=self._cur, mo_out=mo)
                self._ets.append(mo)
            elif k in (1, 2):
                nxt = self._buf(sample)
                out = ops.cfg_pndm_step(_lib.TCX_PNDM_PRK_MID, uncond, cond, self._cur_sample, guidance, [1 / 3] + coef,
                                        cur_in=self._cur, cur_out=nxt)
                self._cur = nxt
            else:
                out = ops.cfg_pndm_step(_lib.TCX_PNDM_PRK_LAST, uncond, cond, self._cur_sample, guidance, [1 / 6] + coef, cur_in=self._cur)
                self._cur = None
        else:                                                            # 4th-order linear multistep (library: step_plms)
            if len(self._ets) < 3:
                raise RuntimeError("PNDM: the multistep phase needs the three Runge-Kutta outputs first")
            coef = self.prev_coeffs(t, t - N // n)
            hist = self._ets[-3:]
            mo = self._ets[0] if len(self._ets) == 4 else self._buf(sample)   # recycle the oldest tensor once four exist
            out = ops.cfg_pndm_step(_lib.TCX_PNDM_PLMS4, uncond, cond, sample, guidance, [1 / 24] + coef,
                                    e1=hist[-1], e2=hist[-2], e3=hist[-3], mo_out=mo)
            self._ets = hist + [mo]
        self._counter += 1
        return out

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = False, **kw):
        """diffusers-shaped step (no guidance) on the fused kernel."""
        return (self.fused_cfg_step(model_output.contiguous(), None, sample.contiguous(), 1.0, timestep),)
