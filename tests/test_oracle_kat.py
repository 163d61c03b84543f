"""Analytic known-answer tests for the restated `diffusers` arithmetic (parity unpinned by the
reference: it holds no tests or vectors for these; see oracle/diffusers_restated.py)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import diffusers_restated as dr
from oracle.pipeline import get_resize_crop_region_for_grid, prepare_rotary, resize_mask
from oracle.prec import Prec

P = Prec("fp32")


def test_timesteps_flip_sin_to_cos():
    e = dr.timesteps_proj(torch.tensor([0, 7]), 8, True, 0)
    assert e.shape == (2, 8)
    torch.testing.assert_close(e[0], torch.tensor([1., 1, 1, 1, 0, 0, 0, 0]))
    f = torch.exp(-math.log(10000) * torch.arange(4) / 4)
    torch.testing.assert_close(e[1, :4], torch.cos(7 * f))
    torch.testing.assert_close(e[1, 4:], torch.sin(7 * f))


def test_layer_norm_zero_chunk_order_and_modulation():
    D, te = 4, 3
    sd = {"linear.weight": torch.zeros(6 * D, te), "linear.bias": torch.arange(6 * D).float(),
          "norm.weight": torch.ones(D), "norm.bias": torch.zeros(D)}
    x = torch.tensor([[[1., 2, 3, 4]]])
    enc = torch.tensor([[[4., 3, 2, 1]]])
    h, e, g, eg = dr.layer_norm_zero(P, sd, "", x, enc, torch.zeros(1, te), 1e-5)
    n = F.layer_norm(x, (D,), eps=1e-5)
    shift, scale, gate = sd["linear.bias"][0:4], sd["linear.bias"][4:8], sd["linear.bias"][8:12]
    torch.testing.assert_close(h, n * (1 + scale) + shift)
    torch.testing.assert_close(g[0, 0], gate)
    ne = F.layer_norm(enc, (D,), eps=1e-5)
    torch.testing.assert_close(e, ne * (1 + sd["linear.bias"][16:20]) + sd["linear.bias"][12:16])
    torch.testing.assert_close(eg[0, 0], sd["linear.bias"][20:24])


def test_ada_layer_norm_shift_first():
    D, te = 4, 2
    sd = {"linear.weight": torch.zeros(2 * D, te), "linear.bias": torch.arange(2 * D).float(),
          "norm.weight": torch.ones(D), "norm.bias": torch.zeros(D)}
    x = torch.tensor([[[1., 2, 3, 5]]])
    y = dr.ada_layer_norm(P, sd, "", x, torch.zeros(1, te), 1e-5)
    n = F.layer_norm(x, (D,), eps=1e-5)
    torch.testing.assert_close(y, n * (1 + torch.arange(4, 8).float()) + torch.arange(0, 4).float())


def test_rope_tables_and_rotation():
    cos, sin = dr.get_3d_rotary_pos_embed(64, ((0, 0), (30, 45)), (30, 45), 13)
    assert cos.shape == (13 * 30 * 45, 64) and cos.dtype == torch.float32
    # token (t=2,h=3,w=5): dims 0:16 temporal, 16:40 height, 40:64 width, each freq repeated twice
    idx = (2 * 30 + 3) * 45 + 5
    ft = 1.0 / (10000 ** (torch.arange(0, 16, 2).float() / 16))
    fh = 1.0 / (10000 ** (torch.arange(0, 24, 2).float() / 24))
    exp = torch.cat([(2 * ft).repeat_interleave(2), (3 * fh).repeat_interleave(2), (5 * fh).repeat_interleave(2)])
    torch.testing.assert_close(cos[idx], exp.cos())
    torch.testing.assert_close(sin[idx], exp.sin())
    # rotation of adjacent pairs: (a,b) -> (a cos - b sin, b cos + a sin)
    x = torch.randn(1, 1, 1, 64)
    y = dr.apply_rotary_emb(x, cos[idx:idx + 1], sin[idx:idx + 1])
    a, b = x[0, 0, 0, 0::2], x[0, 0, 0, 1::2]
    c, s = exp.cos()[0::2], exp.sin()[0::2]
    torch.testing.assert_close(y[0, 0, 0, 0::2], a * c - b * s)
    torch.testing.assert_close(y[0, 0, 0, 1::2], b * c + a * s)
    # norm preserved
    torch.testing.assert_close(y.norm(), x.norm())


def test_rope_crop_region_384x672_is_fractional():
    assert get_resize_crop_region_for_grid((30, 45), 45, 30) == ((0, 0), (30, 45))
    assert get_resize_crop_region_for_grid((24, 42), 45, 30) == ((2, 0), (28, 45))
    cos, _ = prepare_rotary(384, 672, 13, 2, 64)
    assert cos.shape == (13 * 24 * 42, 64)


def test_ddim_zero_terminal_snr_and_trailing_timesteps():
    s = dr.DDIMScheduler()
    assert abs(float(s.alphas_cumprod[-1])) < 1e-10          # zero terminal SNR
    assert abs(float(s.alphas_cumprod[0]) - (1 - 0.00085)) < 1e-6
    s.set_timesteps(50)
    assert s.timesteps[0] == 999 and s.timesteps[-1] == 19 and len(s.timesteps) == 50
    assert torch.all(s.timesteps[:-1] - s.timesteps[1:] == 20)
    # v-prediction at t=999 (abar=0): x0 = -v, eps = sample
    x = torch.randn(1, 2, 3)
    v = torch.randn(1, 2, 3)
    a_prev = s.alphas_cumprod[979]
    out = s.step(P, v, 999, x)
    torch.testing.assert_close(out, a_prev.sqrt() * (-v) + (1 - a_prev).sqrt() * x, rtol=1e-5, atol=1e-6)
    # last step lands on x0 (final alpha = 1)
    a = s.alphas_cumprod[19]
    out = s.step(P, v, 19, x)
    torch.testing.assert_close(out, a.sqrt() * x - (1 - a).sqrt() * v, rtol=1e-5, atol=1e-6)


def test_ddim_bf16_promotion_quirk():
    s = dr.DDIMScheduler()
    s.set_timesteps(50)
    pb = Prec("bf16")
    x = torch.randn(64).to(torch.bfloat16)
    v = torch.randn(64)
    a, ap = s.coeffs(499)
    out = s.step(pb, v, 499, x)
    sa, sb = a.sqrt(), (1 - a).sqrt()
    x0 = (sa * x.float()).bfloat16().float() - sb * v
    eps = sa * v + (sb * x.float()).bfloat16().float()
    torch.testing.assert_close(out, ap.sqrt() * x0 + (1 - ap).sqrt() * eps)


def test_cog_ddim_is_the_same_ddim_update_on_a_shifted_schedule():
    """`CogVideoXDDIMScheduler` ("DDIM_Cog"), known answers: (1) with snr_shift_scale = 1 its float64 schedule equals
    DDIMScheduler's to fp32 precision (both are the zero-terminal-SNR rescale of the scaled-linear schedule; rescaling
    alphas_cumprod directly or through the betas is the same map); (2) the SNR shift divides the SNR a/(1-a) by s exactly;
    (3) its `a x + b x0` step is algebraically the DDIM update sqrt(a_prev) x0 + sqrt(1-a_prev) eps: equal in fp32 to 1e-5;
    (4) first / last step limits."""
    c, d = dr.CogVideoXDDIMScheduler(), dr.DDIMScheduler()
    assert c.alphas_cumprod.dtype == torch.float64
    torch.testing.assert_close(c.alphas_cumprod.float(), d.alphas_cumprod, rtol=2e-4, atol=1e-7)
    assert abs(float(c.alphas_cumprod[-1])) < 1e-20 and abs(float(c.alphas_cumprod[0]) - (1 - 0.00085)) < 1e-9
    raw = torch.cumprod(1 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2, 0)
    sh = dr.CogVideoXDDIMScheduler(snr_shift_scale=3.0, rescale_betas_zero_snr=False).alphas_cumprod
    torch.testing.assert_close(sh / (1 - sh), raw / (1 - raw) / 3.0, rtol=1e-12, atol=0)
    c.set_timesteps(50), d.set_timesteps(50)
    assert c.timesteps.tolist() == d.timesteps.tolist()
    x, v = torch.randn(1, 2, 3), torch.randn(1, 2, 3)
    for t in (999, 979, 499, 19):
        torch.testing.assert_close(c.step(P, v, t, x), d.step(P, v, t, x), rtol=2e-4, atol=2e-5)
    a_prev = c.alphas_cumprod[979]
    torch.testing.assert_close(c.step(P, v, 999, x), (a_prev.sqrt() * (-v) + (1 - a_prev).sqrt() * x).float(), rtol=1e-5, atol=1e-6)
    a = c.alphas_cumprod[19]
    torch.testing.assert_close(c.step(P, v, 19, x), (a.sqrt() * x - (1 - a).sqrt() * v).float(), rtol=1e-5, atol=1e-6)


def test_cog_ddim_bf16_rounding_points():
    c = dr.CogVideoXDDIMScheduler()
    c.set_timesteps(50)
    x = torch.randn(64).to(torch.bfloat16)
    v = torch.randn(64)
    a_t, a_prev = c.coeffs(499)
    sa, sb = float(a_t ** 0.5), float((1 - a_t) ** 0.5)
    ca = float(((1 - a_prev) / (1 - a_t)) ** 0.5)
    cb = float(a_prev ** 0.5 - a_t ** 0.5 * ((1 - a_prev) / (1 - a_t)) ** 0.5)
    x0 = (sa * x.float()).bfloat16().float() - sb * v
    torch.testing.assert_close(c.step(Prec("bf16"), v, 499, x), (ca * x.float()).bfloat16().float() + cb * x0)


def test_upsample3d_frame_rules():
    x = torch.arange(3.)[None, None, :, None, None].expand(1, 1, 3, 2, 2)
    y = dr.upsample3d_nearest(x, True)                 # odd T: frame0 spatial only, rest x2 in time
    assert y.shape == (1, 1, 5, 4, 4)
    assert y[0, 0, :, 0, 0].tolist() == [0, 1, 1, 2, 2]
    y = dr.upsample3d_nearest(x[:, :, :2], True)
    assert y.shape == (1, 1, 4, 4, 4) and y[0, 0, :, 0, 0].tolist() == [0, 0, 1, 1]
    y = dr.upsample3d_nearest(x, False)
    assert y.shape == (1, 1, 3, 4, 4)


def test_sdpa_matches_torch():
    q, k, v = torch.randn(3, 2, 2, 17, 8).unbind(0)
    torch.testing.assert_close(dr.sdpa(P, q, k, v, 8 ** -0.5), F.scaled_dot_product_attention(q, k, v),
                               rtol=1e-5, atol=1e-6)


def test_feed_forward_gelu_tanh():
    sd = {"net.0.proj.weight": torch.eye(3), "net.0.proj.bias": torch.zeros(3),
          "net.2.weight": torch.eye(3), "net.2.bias": torch.zeros(3)}
    x = torch.tensor([[-1.0, 0.0, 2.0]])
    g = 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))
    torch.testing.assert_close(dr.feed_forward(P, sd, "", x), g)


def test_resize_mask_first_frame_separate():
    m = torch.zeros(1, 1, 9, 16, 16)
    m[:, :, 0] = 1
    lat = torch.zeros(1, 16, 3, 2, 2)
    r = resize_mask(m, lat)
    assert r.shape == (1, 1, 3, 2, 2)
    assert torch.all(r[:, :, 0] == 1) and torch.all(r[:, :, 1:] == 0)


def test_image_preprocess_and_mask_binarise():
    x = torch.tensor([0.0, 0.25, 1.0]).view(1, 1, 1, 3)
    torch.testing.assert_close(dr.vae_image_preprocess(x, 1, 3), torch.tensor([-1.0, -0.5, 1.0]).view(1, 1, 1, 3))
    m = torch.tensor([0.0, 0.4, 255.0]).view(1, 1, 1, 3)
    assert dr.vae_image_preprocess(m, 1, 3, do_normalize=False, do_binarize=True).flatten().tolist() == [0, 0, 1]


def test_precision_modes_are_ordered():
    """bf16 contract is at least as close to fp32 as the reference's per-op bf16 rounding."""
    torch.manual_seed(0)
    D, te = 64, 16
    sd = {"linear.weight": torch.randn(6 * D, te) * 0.1, "linear.bias": torch.randn(6 * D) * 0.1,
          "norm.weight": torch.ones(D), "norm.bias": torch.zeros(D)}
    x, e, t = torch.randn(2, 50, D), torch.randn(2, 5, D), torch.randn(2, te)
    ref = dr.layer_norm_zero(Prec("fp32"), sd, "", x, e, t, 1e-5)[0]
    c = dr.layer_norm_zero(Prec("bf16"), sd, "", x.bfloat16().float(), e.bfloat16().float(), t.bfloat16().float(), 1e-5)[0]
    r = dr.layer_norm_zero(Prec("bf16_ref"), sd, "", x.bfloat16().float(), e.bfloat16().float(), t.bfloat16().float(), 1e-5)[0]
    assert (c - ref).abs().mean() <= (r - ref).abs().mean() * 1.05


def test_forward_warp_identity_and_occlusion():
    """Known answers for the point-cloud render (reference models/utils.py:220-293, 422-583):
    same pose -> every pixel lands on itself; a one-pixel shift of a fronto-parallel plane moves the image by one
    column and leaves the uncovered column at -1 / mask 0; of two surfaces landing on one pixel the nearer one wins
    (weight 1/exp(50 log(1+d)/max))."""
    from oracle import warp
    g = torch.Generator().manual_seed(5)
    b, h, w, f = 1, 6, 8, 10.0
    frame = torch.rand(b, 3, h, w, generator=g) * 2 - 1
    depth = torch.full((b, 1, h, w), 2.0)
    k = torch.tensor([[f, 0, 4.0], [0, f, 3.0], [0, 0, 1]])[None]
    eye = torch.eye(4)[None]
    out, mask, wd, flow = warp.forward_warp(frame, None, depth, eye, eye, k)
    assert float(flow.abs().max()) < 1e-5 and torch.equal(mask, torch.ones_like(mask))
    torch.testing.assert_close(out, frame, rtol=0, atol=1e-5)
    torch.testing.assert_close(wd, depth, rtol=0, atol=1e-5)
    t2 = eye.clone()
    t2[0, 0, 3] = 2.0 / f                                      # x' = x + f*tx/z = x + 1
    out, mask, wd, flow = warp.forward_warp(frame, None, depth, eye, t2, k)
    torch.testing.assert_close(flow[:, 0], torch.ones(b, h, w), rtol=0, atol=1e-5)
    torch.testing.assert_close(out[..., 1:], frame[..., :-1], rtol=0, atol=2e-5)
    # uncovered column: only rounding-size weights can land there
    assert float(mask[..., 1:].min()) == 1.0
    near = depth.clone()
    near[..., 2] = 1.0                                         # column 2 is twice as close: shifts by 2, lands on column 4
    out, mask, wd, _ = warp.forward_warp(frame, None, near, eye, t2, k)
    torch.testing.assert_close(out[..., 4], frame[..., 2], rtol=0, atol=1e-4)      # far column 3 also lands there, loses
    torch.testing.assert_close(wd[..., 4], torch.ones(b, 1, h), rtol=0, atol=1e-4)


def test_sincos_position_table_known_answers():
    """diffusers `get_3d_sincos_pos_embed` restated (oracle/diffusers_restated.py; parity unpinned): analytic entries.  Channel
    layout [frame D/4 | first spatial half 3D/8 | second 3D/8], each [sin | cos]; row (t, h, w) with w fastest; frequency i of a
    part of width d is 10000^(-i / (d/2)); positions are divided by the interpolation scales."""
    import numpy as np
    import pytest
    D, W, H, T, ss, ts = 64, 5, 3, 4, 1.875, 2.0
    tab = dr.get_3d_sincos_pos_embed(D, (W, H), T, ss, ts)
    assert tab.shape == (T, H * W, D) and tab.dtype == np.float64
    dt, dsp = D // 4, 3 * D // 4
    half = dsp // 2
    t, h, w = 3, 2, 4
    row = tab[t, h * W + w]
    # frame part: sin | cos of (t / ts) * omega_i, omega_i = 10000^(-i/(dt/2))
    om_t = 1.0 / 10000 ** (np.arange(dt // 2) / (dt / 2))
    np.testing.assert_allclose(row[:dt // 2], np.sin(np.float32(t) / ts * om_t), rtol=0, atol=1e-12)
    np.testing.assert_allclose(row[dt // 2:dt], np.cos(np.float32(t) / ts * om_t), rtol=0, atol=1e-12)
    # spatial part: the first half carries the COLUMN coordinate (meshgrid(w, h)[0]), the second the row coordinate
    om_s = 1.0 / 10000 ** (np.arange(half // 2) / (half / 2))
    cw, ch = np.float32(w) / np.float32(ss), np.float32(h) / np.float32(ss)
    np.testing.assert_allclose(row[dt:dt + half // 2], np.sin(cw * om_s), rtol=0, atol=1e-12)
    np.testing.assert_allclose(row[dt + half // 2:dt + half], np.cos(cw * om_s), rtol=0, atol=1e-12)
    np.testing.assert_allclose(row[dt + half:dt + half + half // 2], np.sin(ch * om_s), rtol=0, atol=1e-12)
    np.testing.assert_allclose(row[dt + half + half // 2:], np.cos(ch * om_s), rtol=0, atol=1e-12)
    # position 0 in every axis: sin = 0, cos = 1
    z = tab[0, 0]
    assert np.all(z[:dt // 2] == 0) and np.all(z[dt // 2:dt] == 1) and np.all(z[dt:dt + half // 2] == 0)
    with pytest.raises(ValueError):
        dr.get_3d_sincos_pos_embed(66, (2, 2), 1)


def test_sigma_sampler_tables_known_answers():
    """"Euler" / "Euler A" / "DPM++" restated (oracle/diffusers_restated.py; parity unpinned): the sigma table.  sigma = sqrt((1 - abar)
    / abar) of the zero-SNR-rescaled scaled-linear schedule with abar_T pinned to 2^-24 -> sigma_max = sqrt(2^24 - 1); trailing
    timesteps 999, 979, ..., 19; a final sigma of 0; Euler starts from noise * sigma_max and feeds x / sqrt(sigma^2 + 1) to the model;
    DPM++ starts from unit noise and uses int64 timesteps."""
    import numpy as np
    for cls in (dr.EulerDiscreteScheduler, dr.EulerAncestralDiscreteScheduler, dr.DPMSolverMultistepScheduler):
        s = cls()
        s.set_timesteps(50)
        assert s.timesteps.tolist() == list(range(999, 0, -20)) and len(s.sigmas) == 51 and float(s.sigmas[-1]) == 0.0
        assert float(s.sigmas[0]) == float(np.float32(np.sqrt(np.float32(1 - 2.0 ** -24) / np.float32(2.0 ** -24))))
        assert bool((s.sigmas[:-1] > s.sigmas[1:]).all())
        abar = dr.DDIMScheduler().alphas_cumprod                       # the same schedule below the last entry
        np.testing.assert_allclose(float(s.sigmas[1]), float(((1 - abar[979]) / abar[979]) ** 0.5), rtol=1e-6)
    e, d = dr.EulerDiscreteScheduler(), dr.DPMSolverMultistepScheduler()
    e.set_timesteps(50), d.set_timesteps(50)
    assert e.timesteps.dtype == torch.float32 and d.timesteps.dtype == torch.int64
    assert float(e.init_noise_sigma) == float(e.sigmas[0]) and d.init_noise_sigma == 1.0
    x = torch.randn(3, 4, generator=torch.Generator().manual_seed(0))
    torch.testing.assert_close(e.scale_model_input(Prec("fp32"), x, 979.0), x / (float(e.sigmas[1]) ** 2 + 1) ** 0.5, rtol=1e-6, atol=0)


def test_euler_step_known_answers():
    """One Euler step of the probability-flow ODE dx/dsigma = (x - x0) / sigma with an exact v-prediction moves x on the straight
    line to x0: x_next = x0 + (sigma_next / sigma)(x - x0) (independent float64 evaluation).  "Euler A": sigma_up^2 + sigma_down^2 =
    sigma_next^2, zero noise = the Euler step to sigma_down, the noise enters with weight sigma_up; the last step lands on x0."""
    g = torch.Generator().manual_seed(1)
    x0, eps = torch.randn(5, 7, generator=g), torch.randn(5, 7, generator=g)
    p = Prec("fp32")
    for cls in (dr.EulerDiscreteScheduler, dr.EulerAncestralDiscreteScheduler):
        s = cls()
        s.set_timesteps(50)
        for i in (0, 1, 25, 48, 49):
            t, sg, sg_to = s.timesteps[i], float(s.sigmas[i]), float(s.sigmas[i + 1])
            x = x0 + sg * eps                                          # the sample at noise level sigma (sigma parametrisation)
            a = 1 / (sg ** 2 + 1) ** 0.5
            v = a * eps - sg * a * x0                                  # exact v for the scaled input a x = a x0 + (sg a) eps
            nz = torch.randn(5, 7, generator=g)
            got = s.step(p, v.float(), t, x, noise=nz) if s.ancestral else s.step(p, v.float(), t, x)
            if not s.ancestral:
                want = x0.double() + (sg_to / sg) * (x.double() - x0.double())
            else:
                up = (sg_to ** 2 * (sg ** 2 - sg_to ** 2) / sg ** 2) ** 0.5
                down = (sg_to ** 2 - up ** 2) ** 0.5
                assert abs(up ** 2 + down ** 2 - sg_to ** 2) <= 1e-9 * max(sg_to ** 2, 1e-30)
                want = x0.double() + (down / sg) * (x.double() - x0.double()) + up * nz.double()
                torch.testing.assert_close(s.step(p, v.float(), t, x, noise=torch.zeros(5, 7)),
                                           s.step(p, v.float(), t, x, noise=nz) - nz * s.step_coeffs(t)[4], rtol=1e-5, atol=1e-5 * max(sg, 1.0))
            torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-4 * max(sg, 1.0) * 1e-2)
            if i == 49:
                torch.testing.assert_close(got, x0, rtol=1e-5, atol=1e-5)


def test_dpmpp_known_answers():
    """DPM-Solver++: the first-order update IS the DDIM step in the alpha / sigma parametrisation (x_next = alpha_next x0 + sig_next eps
    with x0, eps from the v-prediction), checked against an independent float64 DDIM evaluation; the second-order (2M, midpoint)
    update against its closed form x_next = A x - B (x0_i + (x0_i - x0_{i-1}) / (2 r0)); order 1 on the first and on the last step."""
    import math
    g = torch.Generator().manual_seed(2)
    p = Prec("fp32")
    s = dr.DPMSolverMultistepScheduler()
    s.set_timesteps(50)
    al = lambda sg: 1 / math.sqrt(sg ** 2 + 1)
    x, v = torch.randn(4, 6, generator=g), torch.randn(4, 6, generator=g)
    # step 0: first order
    sg0, sg1, sg2 = (float(s.sigmas[i]) for i in (0, 1, 2))
    a0, a1, a2 = al(sg0), al(sg1), al(sg2)
    b0, b1, b2 = sg0 * a0, sg1 * a1, sg2 * a2
    assert s.step_coeffs(999)[4] is None
    got = s.step(p, v, 999, x)
    x0_0 = a0 * x.double() - b0 * v.double()
    eps_0 = a0 * v.double() + b0 * x.double()
    torch.testing.assert_close(got.double(), a1 * x0_0 + b1 * eps_0, rtol=1e-4, atol=1e-4)          # == DDIM
    # step 1: second order with the history of step 0
    assert s.lower_order_nums == 1 and s.step_coeffs(979)[4] is not None
    v1 = torch.randn(4, 6, generator=g)
    got2 = s.step(p, v1, 979, got)
    x0_1 = a1 * got.double() - b1 * v1.double()
    lam = lambda a, b: math.log(a) - math.log(b)
    h, h0 = lam(a2, b2) - lam(a1, b1), lam(a1, b1) - lam(a0, b0)
    A, B = b2 / b1, a2 * (math.exp(-h) - 1.0)
    want2 = A * got.double() - B * (x0_1 + (x0_1 - x0_0) / (2 * (h0 / h)))
    torch.testing.assert_close(got2.double(), want2, rtol=1e-4, atol=1e-4)
    # last step: first order onto x0 (final sigma 0: A = 0, B = -1)
    s.lower_order_nums = 2
    c = s.step_coeffs(19)
    assert c[4] is None and float(c[2]) == 0.0 and float(c[3]) == -1.0


def test_pndm_known_answers():
    """"PNDM" restated (oracle/diffusers_restated.py; parity unpinned): (1) the schedule — 59 evaluations for 50 steps, the first 12 a
    classical Runge-Kutta pattern t, t - h/2, t - h/2, t - h over three intervals of h = 20; (2) the PNDM transfer formula
    (`_get_prev_sample`) with an exact v-prediction IS the DDIM step sqrt(a_prev) x0 + sqrt(1 - a_prev) eps (independent float64
    evaluation), also onto final alpha 1; (3) a constant model output comes out of the Runge-Kutta combination (1/6 + 1/3 + 1/3 + 1/6)
    and of the multistep combination ((55 - 59 + 37 - 9) / 24) unchanged."""
    s = dr.PNDMScheduler()
    s.set_timesteps(50)
    ts = s.timesteps.tolist()
    assert len(ts) == 59 and ts[:12] == [999, 989, 989, 979, 979, 969, 969, 959, 959, 949, 949, 939] and ts[12:15] == [939, 919, 899] and ts[-1] == 19
    p = Prec("fp32")
    g = torch.Generator().manual_seed(3)
    x0, eps = torch.randn(5, 6, generator=g).double(), torch.randn(5, 6, generator=g).double()
    for t, prev in ((999, 979), (519, 499), (19, -1), (999, 989)):
        a = s.alphas_cumprod[t].double()
        ap = (s.alphas_cumprod[prev] if prev >= 0 else s.final_alpha_cumprod).double()
        x, v = a.sqrt() * x0 + (1 - a).sqrt() * eps, a.sqrt() * eps - (1 - a).sqrt() * x0
        got = s._get_prev_sample(p, x.float(), t, prev, v.float()).double()
        torch.testing.assert_close(got, ap.sqrt() * x0 + (1 - ap).sqrt() * eps, rtol=1e-5, atol=1e-5)
    # constant model output through one Runge-Kutta group and one multistep update
    s.set_timesteps(50)
    x = torch.randn(5, 6, generator=g)
    v = torch.randn(5, 6, generator=g)
    outs = [s.step(p, v, t, x) for t in ts[:4]]
    want = s._get_prev_sample(p, x, 999, 979, v)
    torch.testing.assert_close(outs[3], want, rtol=1e-6, atol=1e-6)             # the 4th evaluation lands where a single step with v would
    torch.testing.assert_close(outs[0], s._get_prev_sample(p, x, 999, 989, v), rtol=0, atol=0)
    for t in ts[4:12]:
        s.step(p, v, t, x)
    assert len(s.ets) == 3 and s.counter == 12
    y = s.step(p, v, ts[12], x)
    torch.testing.assert_close(y, s._get_prev_sample(p, x, 939, 919, v), rtol=1e-6, atol=1e-6)
    assert len(s.ets) == 4
