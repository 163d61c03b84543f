"""HIP kernels (through the C ABI) vs the CPU oracle on identical seeded inputs.  GPU only.

Tolerances (written where used):
  * fp32-output attention: rtol 1e-3 / atol 1e-4 against the oracle contract (north_star tolerance);
  * bf16 outputs: within one bf16 ulp (rtol 2^-7) + atol 1e-3 of the oracle's fp32 value, and the
    mean abs error must be below a quarter ulp of the output scale (catches systematic error).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import diffusers_restated as dr
from oracle import transformer as otr
from oracle import vae as ovae
from oracle.prec import Prec

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from trajectorycrafter_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def bf(t):
    return t.to(BF)


def assert_bf16_close(got, ref, atol=1e-3, ulps=1.0, mean_frac=0.25, extra=0.0):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    bound = ref.abs() * (2.0 ** -7) * ulps + atol + extra
    bad = err > bound
    assert not bad.any(), f"{int(bad.sum())}/{bad.numel()} outside 1 bf16 ulp; max err {float(err.max()):.4g}"
    scale = float(ref.abs().mean()) + 1e-6
    assert float(err.mean()) <= mean_frac * scale * 2.0 ** -7 + atol * 0.1, (float(err.mean()), scale)


def oracle_attn(q, k, v, scale, with_bound=False):
    """q,k,v [B,S,H,D] bf16 -> fp32 [B,Sq,H,D] under the bf16 contract (unnormalised P rounded before PV).

    `with_bound` also returns the rigorous P-rounding bound 2^-8 * softmax(S) @ |V|: kernel and oracle each
    round every probability once (relative error <= 2^-9 each) but against different running maxima."""
    p = Prec("bf16")
    qt, kt, vt = q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)
    o = dr.sdpa(p, qt, kt, vt, scale).transpose(1, 2).contiguous()
    if not with_bound:
        return o
    pr = torch.softmax(torch.matmul(qt, kt.transpose(-1, -2)) * scale, dim=-1)
    return o, (2.0 ** -8) * torch.matmul(pr, vt.abs()).transpose(1, 2).contiguous()


def assert_attn_close(got, ref, bound, rtol=1e-3, atol=1e-4):
    """north_star tolerance (rtol 1e-3 / atol 1e-4) + the P-rounding bound; mean error an order below it."""
    got = got.float().cpu()
    err = (got - ref).abs()
    lim = rtol * ref.abs() + atol + bound
    assert torch.isfinite(got).all()
    assert not (err > lim).any(), f"max err {float(err.max()):.3g}, worst excess {float((err - lim).max()):.3g}"
    assert float(err.mean()) < 0.15 * float(lim.mean()), (float(err.mean()), float(lim.mean()))


@pytest.mark.parametrize("D,B,H,Sq,Sk", [(64, 1, 2, 300, 333), (64, 2, 3, 513, 64), (128, 1, 2, 257, 200),
                                         (128, 2, 1, 64, 450), (64, 1, 1, 1, 1), (64, 1, 1, 50, 128), (64, 1, 1, 50, 320),
                                         (128, 1, 1, 30, 65), (128, 1, 1, 30, 192)])
def test_attn_fwd_matches_oracle(ops, D, B, H, Sq, Sk):
    g = torch.Generator().manual_seed(D + Sq)
    q, k, v = (bf(torch.randn(B, s, H, D, generator=g)) for s in (Sq, Sk, Sk))
    scale = D ** -0.5
    ref, bound = oracle_attn(q, k, v, scale, with_bound=True)
    o32 = ops.attn_fwd(dev(q), dev(k), dev(v), scale, out_dtype=torch.float32)
    assert_attn_close(o32, ref, bound)            # fp32 output mode
    o16 = ops.attn_fwd(dev(q), dev(k), dev(v), scale)
    assert o16.dtype == BF
    assert_bf16_close(o16, ref, extra=bound)


@pytest.mark.parametrize("B,H,Sq,Sk", [(1, 2, 300, 333), (2, 3, 513, 64), (1, 1, 1, 1), (1, 2, 70, 129), (1, 1, 256, 2048),
                                       (1, 1, 40, 128), (1, 2, 33, 100), (1, 1, 64, 300), (1, 1, 64, 449)])   # 2, 2, 5, 8 key tiles
@pytest.mark.parametrize("D", [64, 128])
def test_attn_fwd_log2_scores_fast_path(ops, B, H, Sq, Sk, D):
    """TCX_ATTN_LOG2_SCORES (D = 64 and 128): q pre-multiplied by scale*log2(e); running max as the MFMA initial
    accumulator.  Oracle: dr.sdpa_log2."""
    g = torch.Generator().manual_seed(Sq + Sk + D)
    q, k, v = (bf(torch.randn(B, s, H, D, generator=g)) for s in (Sq, Sk, Sk))
    q = bf(q.float() * (D ** -0.5 * 1.4426950408889634))
    qt, kt, vt = q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)
    ref = dr.sdpa_log2(Prec("bf16"), qt, kt, vt).transpose(1, 2).contiguous()
    pr = torch.softmax(torch.matmul(qt, kt.transpose(-1, -2)) * math.log(2.0), dim=-1)
    # P is rounded once on each side, and here the rounded P also feeds the row sum: 3 * 2^-9
    bound = 3 * (2.0 ** -9) * torch.matmul(pr, vt.abs()).transpose(1, 2).contiguous()
    o32 = ops.attn_fwd(dev(q), dev(k), dev(v), 1.0, out_dtype=torch.float32, log2_scores=True)
    assert_attn_close(o32, ref, bound)
    o16 = ops.attn_fwd(dev(q), dev(k), dev(v), 1.0, log2_scores=True)
    assert_bf16_close(o16, ref, extra=bound)
    with pytest.raises(ops.TcxError):
        ops.attn_fwd(dev(q), dev(k), dev(v), 0.125, log2_scores=True)       # scale must be 1 with the flag
    # bound-centred loop (softmax centred on |q| max|k| from k_sqmax): same contract, same tolerance
    ksq = dev((k.float() ** 2).sum(-1).amax(1).contiguous())                # [B, H]
    o32b = ops.attn_fwd(dev(q), dev(k), dev(v), 1.0, out_dtype=torch.float32, log2_scores=True, k_sqmax=ksq)
    assert_attn_close(o32b, ref, bound)
    o16b = ops.attn_fwd(dev(q), dev(k), dev(v), 1.0, log2_scores=True, k_sqmax=ksq)
    assert_bf16_close(o16b, ref, extra=bound)


def test_attn_fwd_fast_path_forced_recentre(ops):
    """Rule 26 for the FAST path: scores that start very negative (first-tile re-centre with delta < 0), a
    late spike above the deferral threshold for some rows only, and a ragged last tile."""
    g = torch.Generator().manual_seed(17)
    B, H, S, D = 1, 1, 64 * 5 + 9, 64
    q, k, v = (bf(torch.randn(B, S, H, D, generator=g)) for _ in range(3))
    q = bf(q.float() * 0.18)
    k[0, :64, 0] = -q[0, 3, 0] * 40.0                 # row 3: every key of tile 0 scores hugely negative
    k[0, 64 * 3 + 7, 0] = q[0, 10, 0] * 60.0          # row 10 jumps far beyond 2^6 at tile 3
    k[0, 64 * 5 + 8, 0] = q[0, 40, 0] * 90.0          # row 40 jumps at the (ragged) last tile
    qt, kt, vt = q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)
    ref = dr.sdpa_log2(Prec("bf16"), qt, kt, vt).transpose(1, 2).contiguous()
    pr = torch.softmax(torch.matmul(qt, kt.transpose(-1, -2)) * math.log(2.0), dim=-1)
    bound = 3 * (2.0 ** -9) * torch.matmul(pr, vt.abs()).transpose(1, 2).contiguous()
    o = ops.attn_fwd(dev(q), dev(k), dev(v), 1.0, out_dtype=torch.float32, log2_scores=True)
    assert_attn_close(o, ref, bound)
    # with k_sqmax: the spiked keys push the Cauchy-Schwarz bound of these rows beyond 60 -> the workgroup is
    # handed to the exact kernel (complementary launch); result must be the same
    ksq = dev((k.float() ** 2).sum(-1).amax(1).contiguous())
    assert float(ksq.max()) * float((q.float() ** 2).sum(-1).max()) > 60.0 ** 2
    ob = ops.attn_fwd(dev(q), dev(k), dev(v), 1.0, out_dtype=torch.float32, log2_scores=True, k_sqmax=ksq)
    assert_attn_close(ob, ref, bound)


@pytest.mark.parametrize("D", [64, 128])
def test_attn_fwd_bound_mixed_workgroups(ops, D):
    """Some 256-row workgroups safe (bound loop), one with a huge-norm query row (exact loop): every row correct."""
    g = torch.Generator().manual_seed(23 + D)
    B, H, S = 1, 2, 700
    q, k, v = (bf(torch.randn(B, S, H, D, generator=g)) for _ in range(3))
    q = bf(q.float() * (0.18 if D == 64 else 0.12))
    q[0, 300, 1] *= 40.0                              # M = |q||k|max ~ 1.4*40*10 >> 60 for the workgroup of rows 256..511, head 1
    qt, kt, vt = q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)
    ref = dr.sdpa_log2(Prec("bf16"), qt, kt, vt).transpose(1, 2).contiguous()
    pr = torch.softmax(torch.matmul(qt, kt.transpose(-1, -2)) * math.log(2.0), dim=-1)
    bound = 3 * (2.0 ** -9) * torch.matmul(pr, vt.abs()).transpose(1, 2).contiguous()
    dk = dev(k)
    ksq = dev((k.float() ** 2).sum(-1).amax(1).contiguous())
    o = ops.attn_fwd(dev(q), dk, dev(v), 1.0, out_dtype=torch.float32, log2_scores=True, k_sqmax=ksq)
    assert_attn_close(o, ref, bound)


def test_attn_fwd_strided_fused_qkv_views(ops):
    """q/k/v as views into one fused [B,S,3*H*D] projection buffer; o into a strided buffer."""
    g = torch.Generator().manual_seed(5)
    B, S, H, D = 2, 200, 3, 64
    qkv = bf(torch.randn(B, S, 3 * H * D, generator=g))
    ref = oracle_attn(*(t.reshape(B, S, H, D) for t in qkv.chunk(3, -1)), 0.125)
    dq = dev(qkv)
    q, k, v = (t.view(B, S, H, D) for t in dq.chunk(3, -1))
    out = torch.zeros(B, S, H, D, device="cuda", dtype=BF)
    ops.attn_fwd(q, k, v, 0.125, out=out)
    assert_bf16_close(out, ref)


def test_attn_fwd_forced_rescale_branch(ops):
    """Rule 26: spike one key late in the sequence so the running max jumps in a later tile for some
    rows only; exercises the (rare) O-rescale branch and the skip path in the same wave."""
    g = torch.Generator().manual_seed(9)
    B, H, S, D = 1, 1, 64 * 5, 64
    q, k, v = (bf(torch.randn(B, S, H, D, generator=g)) for _ in range(3))
    k[0, 64 * 3 + 7, 0] = q[0, 10, 0] * 6.0          # row 10's max jumps at tile 3
    k[0, 64 * 4 + 1, 0] = q[0, 40, 0] * 9.0          # row 40's max jumps at tile 4
    ref, bound = oracle_attn(q, k, v, 0.125, with_bound=True)
    o = ops.attn_fwd(dev(q), dev(k), dev(v), 0.125, out_dtype=torch.float32)
    assert_attn_close(o, ref, bound)


def test_attn_fwd_rejects_bad_arguments(ops):
    q = torch.zeros(1, 8, 1, 32, device="cuda", dtype=BF)
    with pytest.raises(ops.TcxError):
        ops.attn_fwd(q, q, q, 1.0)                    # head dim 32 unsupported
    q = torch.zeros(1, 8, 1, 64, device="cuda", dtype=torch.float16)
    with pytest.raises(ops.TcxError):
        ops.attn_fwd(q, q, q, 1.0)


@pytest.mark.parametrize("C,rows,text_len", [(3072, 37, 5), (128, 50, 10), (2048, 9, 0), (256, 70, 70),
                                              # >= 4096 rows: the row-looping kernel (parameters in registers)
                                              (3072, 2101, 226), (1024, 2050, 0), (2048, 2049, 2049), (520, 2060, 3)])
def test_layernorm_modulate(ops, C, rows, text_len):
    g = torch.Generator().manual_seed(C)
    B = 2
    x = bf(torch.randn(B, rows, C, generator=g) * 2 + 0.5)
    gamma, beta = bf(1 + 0.1 * torch.randn(C, generator=g)), bf(0.1 * torch.randn(C, generator=g))
    mod = bf(0.3 * torch.randn(B, 6 * C, generator=g))
    shift, scale, _, e_shift, e_scale, _ = mod.chunk(6, dim=1)
    n = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5)
    sc = torch.where(torch.arange(rows)[None, :, None] < text_len, e_scale.float()[:, None], scale.float()[:, None])
    sh = torch.where(torch.arange(rows)[None, :, None] < text_len, e_shift.float()[:, None], shift.float()[:, None])
    ref = n * (1 + sc) + sh
    dm = dev(mod)
    dshift, dscale, _, de_shift, de_scale, _ = dm.chunk(6, dim=1)
    y = ops.layernorm_modulate(dev(x), dev(gamma), dev(beta), 1e-5, dshift, dscale, de_shift, de_scale, text_len)
    assert_bf16_close(y, ref)
    y2 = ops.layernorm_modulate(dev(x), dev(gamma), dev(beta), 1e-5)       # plain LayerNorm
    assert_bf16_close(y2, n)


def test_layernorm_modulate_strided_video_rows(ops):
    """video rows of the joint [B, text+video, C] buffer (batch stride != rows*C)."""
    g = torch.Generator().manual_seed(3)
    for B, S, C, tl in ((2, 40, 256, 8), (2, 2100, 512, 50)):          # second case: row-looping kernel
        x = bf(torch.randn(B, S, C, generator=g))
        gamma, beta = bf(torch.randn(C, generator=g)), bf(torch.randn(C, generator=g))
        ref = F.layer_norm(x[:, tl:].float(), (C,), gamma.float(), beta.float(), 1e-5)
        y = ops.layernorm_modulate(dev(x)[:, tl:], dev(gamma), dev(beta), 1e-5)
        assert_bf16_close(y, ref)


@pytest.mark.parametrize("B,S,H,text_len", [(2, 50, 2, 10), (1, 33, 48, 7), (1, 20, 3, 0), (2, 77, 48, 9), (1, 300, 48, 0)])
def test_qk_layernorm_rope(ops, B, S, H, text_len):
    g = torch.Generator().manual_seed(S)
    D = 64
    qkv = bf(torch.randn(B, S, 3 * H * D, generator=g) * 1.5)
    gq, bq, gk, bk = (bf(1 + 0.2 * torch.randn(D, generator=g)) for _ in range(4))
    ang = torch.rand(S - text_len, D // 2, generator=g) * 6.28
    cos, sin = ang.cos().repeat_interleave(2, -1).contiguous(), ang.sin().repeat_interleave(2, -1).contiguous()
    q, k, v = (t.reshape(B, S, H, D) for t in qkv.chunk(3, -1))

    def ref_one(x, gm, bt):
        n = F.layer_norm(x.float(), (D,), gm.float(), bt.float(), 1e-6).transpose(1, 2)      # [B,H,S,D]
        r = dr.apply_rotary_emb(n[:, :, text_len:], cos, sin)
        return torch.cat([n[:, :, :text_len], r], dim=2).transpose(1, 2)

    rq, rk = ref_one(q, gq, bq), ref_one(k, gk, bk)
    d = dev(qkv)
    dq, dk, dv = (t.view(B, S, H, D) for t in d.chunk(3, -1))
    ops.qk_layernorm_rope(dq, dk, dev(gq), dev(bq), dev(gk), dev(bk), dev(cos), dev(sin), text_len, 1e-6)
    assert_bf16_close(dq, rq)
    assert_bf16_close(dk, rk)
    assert torch.equal(dv.cpu(), v)                      # v untouched
    # q_scale: q (only) leaves pre-multiplied, one rounding
    d = dev(qkv)
    dq, dk, dv = (t.view(B, S, H, D) for t in d.chunk(3, -1))
    ksq = ops.qk_layernorm_rope(dq, dk, dev(gq), dev(bq), dev(gk), dev(bk), dev(cos), dev(sin), text_len, 1e-6,
                                q_scale=0.18033688, want_k_sqmax=True)
    assert_bf16_close(dq, rq * 0.18033688)
    assert_bf16_close(dk, rk)
    # k_sqmax = max over tokens of |k|^2 of the stored (rounded) keys, per (batch, head): exact up to fp32 summation order
    torch.testing.assert_close(ksq.cpu(), (dk.float().cpu() ** 2).sum(-1).amax(1), rtol=1e-5, atol=1e-5)


def test_gated_residual_and_plain_residual(ops):
    g = torch.Generator().manual_seed(1)
    B, rows, C, tl = 2, 30, 256, 6
    x, y = bf(torch.randn(B, rows, C, generator=g)), bf(torch.randn(B, rows, C, generator=g))
    gate = bf(torch.randn(B, 2 * C, generator=g))
    gv, gt = gate.chunk(2, 1)
    gsel = torch.where(torch.arange(rows)[None, :, None] < tl, gt.float()[:, None], gv.float()[:, None])
    ref = x.float() + gsel * y.float()
    dx, dg = dev(x), dev(gate)
    ops.gated_residual_(dx, dev(y), *dg.chunk(2, 1), text_len=tl)
    assert_bf16_close(dx, ref)
    dx = dev(x)
    ops.gated_residual_(dx, dev(y))
    assert_bf16_close(dx, x.float() + y.float())


def test_bias_gelu_tanh_scale_silu(ops):
    g = torch.Generator().manual_seed(2)
    x = bf(torch.randn(37, 512, generator=g) * 3)
    b = bf(torch.randn(512, generator=g))
    ref = F.gelu(x.float() + b.float(), approximate="tanh")
    assert_bf16_close(ops.bias_gelu_tanh_(dev(x), dev(b)), ref)
    assert_bf16_close(ops.bias_gelu_tanh_(dev(x)), F.gelu(x.float(), approximate="tanh"))
    s = 128 ** -0.25
    assert torch.equal(ops.scale_bf16(dev(x), s).cpu(), (x.float() * s).to(BF))       # bit exact: one fp32 mul, one rounding
    assert_bf16_close(ops.silu(dev(x)), F.silu(x.float()))
    odd = bf(torch.randn(1003, generator=g))
    assert_bf16_close(ops.silu(dev(odd)), F.silu(odd.float()))


@pytest.mark.parametrize("D,H", [(128, 16), (64, 3)])
def test_scale_sqmax(ops, D, H):
    """k * scale of the cross-attention fused with max_s |k|^2 per (batch, head); input is the k half of a [B,S,2HD] buffer."""
    g = torch.Generator().manual_seed(D + H)
    B, S, s_ = 2, 77, 0.2973
    kv = bf(torch.randn(B, S, 2 * H * D, generator=g) * 3)
    kv[1, 40, 5 * D // 2: 5 * D // 2 + 8] = 50.0                       # a spike decides one head's maximum
    y, sq = ops.scale_sqmax(dev(kv)[..., :H * D], s_, H, D)
    want = (kv[..., :H * D].float() * s_).to(BF)
    assert torch.equal(y.cpu(), want)                                 # one fp32 multiply, one rounding: bit exact
    ref = (want.float() ** 2).view(B, S, H, D).sum(-1).amax(1)
    torch.testing.assert_close(sq.cpu(), ref, rtol=1e-5, atol=0)      # exact up to fp32 summation order


def test_patchify_unpatchify_bit_exact(ops):
    g = torch.Generator().manual_seed(4)
    B, Fr, H, W, p = 2, 3, 8, 12, 2
    a, b = bf(torch.randn(B, Fr, 16, H, W, generator=g)), bf(torch.randn(B, Fr, 17, H, W, generator=g))
    x = torch.cat([a, b], 2)
    ref = F.unfold(x.reshape(B * Fr, 33, H, W).float(), kernel_size=p, stride=p).transpose(1, 2).reshape(-1, 33 * p * p)
    got = ops.patchify(dev(a), dev(b), p)
    assert torch.equal(got.float().cpu(), ref)
    padded = ops.patchify(dev(a), dev(b), p, k_pad=128).float().cpu()       # rows zero-filled to K = 256 for the GEMM
    assert padded.shape == (ref.shape[0], 256) and torch.equal(padded[:, :132], ref) and float(padded[:, 132:].abs().max()) == 0
    tok = bf(torch.randn(B, Fr * (H // p) * (W // p), 16 * p * p, generator=g))
    r = tok.reshape(B, Fr, H // p, W // p, 16, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    got = ops.unpatchify(dev(tok), B, Fr, 16, H, W, p)
    assert torch.equal(got.cpu(), r)
    got32 = ops.unpatchify(dev(tok), B, Fr, 16, H, W, p, out_dtype=torch.float32)
    assert torch.equal(got32.cpu(), r.float())


@pytest.mark.parametrize("t", [999, 499, 19])
def test_cfg_ddim_step_matches_oracle(ops, t):
    g = torch.Generator().manual_seed(t)
    s = dr.DDIMScheduler()
    s.set_timesteps(50)
    x = bf(torch.randn(1, 13, 16, 12, 18, generator=g))
    pred = torch.randn(2, 13, 16, 12, 18, generator=g)
    u, c = pred.chunk(2)
    ref = s.step(Prec("bf16"), u + 6.0 * (c - u), t, x)
    a_t, a_prev = s.coeffs(t)
    d = dev(pred)
    got = ops.cfg_ddim_step(d[:1], d[1:], dev(x), 6.0, float(a_t), float(a_prev))
    assert_bf16_close(got, ref, atol=1e-5)
    # bf16 predictions, no guidance
    got = ops.cfg_ddim_step(dev(bf(u)), None, dev(x), 1.0, float(a_t), float(a_prev))
    assert_bf16_close(got, s.step(Prec("bf16"), bf(u).float(), t, x), atol=1e-5)


@pytest.mark.parametrize("t", [999, 499, 19])
def test_cfg_ddim_eta_step_matches_oracle(ops, t):
    """`tcx_cfg_ddim_eta_step` (stochastic DDIM, eta > 0) through the product scheduler against the oracle's step with the same variance
    noise; eta = 1 at the last step (final alpha 1: std = 0, the noise has no weight)."""
    from trajectorycrafter_amd.scheduler import DDIMScheduler
    g = torch.Generator().manual_seed(300 + t)
    s, ps = dr.DDIMScheduler(), DDIMScheduler()
    s.set_timesteps(50), ps.set_timesteps(50)
    x = bf(torch.randn(1, 13, 16, 12, 18, generator=g))
    pred = torch.randn(2, 13, 16, 12, 18, generator=g)
    u, c = pred.chunk(2)
    for eta in (0.3, 1.0):
        gd, gr = torch.Generator(device="cuda").manual_seed(11), torch.Generator(device="cuda").manual_seed(11)
        nz = torch.randn(x.shape, generator=gr, device="cuda", dtype=torch.float32).cpu()
        ref = s.step(Prec("bf16"), u + 6.0 * (c - u), t, x, eta=eta, variance_noise=nz)
        d = dev(pred)
        got = ps.fused_cfg_step(d[:1], d[1:], dev(x), 6.0, t, generator=gd, eta=eta)
        assert_bf16_close(got, ref, atol=1e-5)
        assert not torch.equal(got, ps.fused_cfg_step(d[:1], d[1:], dev(x), 6.0, t)) or t == 19
    assert ps.eta_coeffs(19, 1.0)[4] == 0.0


@pytest.mark.parametrize("t", [999, 499, 19])
def test_cfg_ddim_cog_step_matches_oracle(ops, t):
    """`tcx_cfg_ddim_cog_step` (sampler "DDIM_Cog") vs the oracle's CogVideoXDDIMScheduler.step under the bf16 contract, through
    the product scheduler's `fused_cfg_step`."""
    from trajectorycrafter_amd.scheduler import CogVideoXDDIMScheduler
    g = torch.Generator().manual_seed(100 + t)
    s, ps = dr.CogVideoXDDIMScheduler(), CogVideoXDDIMScheduler()
    s.set_timesteps(50), ps.set_timesteps(50)
    x = bf(torch.randn(1, 13, 16, 12, 18, generator=g))
    pred = torch.randn(2, 13, 16, 12, 18, generator=g)
    u, c = pred.chunk(2)
    ref = s.step(Prec("bf16"), u + 6.0 * (c - u), t, x)
    d = dev(pred)
    got = ps.fused_cfg_step(d[:1], d[1:], dev(x), 6.0, t)
    assert_bf16_close(got, ref, atol=1e-5)
    got = ps.fused_cfg_step(dev(bf(u)), None, dev(x), 1.0, t)
    assert_bf16_close(got, s.step(Prec("bf16"), bf(u).float(), t, x), atol=1e-5)
    assert_bf16_close(ps.step(dev(bf(u)), t, dev(x))[0], s.step(Prec("bf16"), bf(u).float(), t, x), atol=1e-5)


@pytest.mark.parametrize("name", ["Euler", "Euler A", "DPM++"])
def test_cfg_sigma_step_matches_oracle(ops, name):
    """`tcx_cfg_sigma_step` (samplers "Euler", "Euler A", "DPM++") through the product schedulers' `fused_cfg_step`, a whole
    5-step trajectory with CFG 6 on fixed random model outputs, against the oracle's restated diffusers steps under the bf16
    contract: the kernel applies the library's per-element fp32 operations in their order, so the latents agree to the bit after
    every step (fp32 predictions); `scale_model_input` = `tcx_div_bf16`, bit-equal as well.  bf16 predictions and the no-guidance
    form (`step`) ride along."""
    from trajectorycrafter_amd import scheduler as S
    pc, oc = {"Euler": (S.EulerDiscreteScheduler, dr.EulerDiscreteScheduler), "Euler A": (S.EulerAncestralDiscreteScheduler,
              dr.EulerAncestralDiscreteScheduler), "DPM++": (S.DPMSolverMultistepScheduler, dr.DPMSolverMultistepScheduler)}[name]
    g = torch.Generator().manual_seed(77)
    p = Prec("bf16")
    for n_steps, pred_bf16 in ((5, False), (50, False), (3, True)):
        ps, s = pc(), oc()
        ps.set_timesteps(n_steps), s.set_timesteps(n_steps)
        x = bf(torch.randn(1, 5, 16, 6, 10, generator=g) * float(s.init_noise_sigma))
        xd = dev(x)
        gen = torch.Generator(device="cuda").manual_seed(5)
        gen_ref = torch.Generator(device="cuda").manual_seed(5)
        for i, t in enumerate(s.timesteps[:6]):
            # scale_model_input
            want_in = s.scale_model_input(p, x.float(), t)
            got_in = ps.scale_model_input(xd, t)
            assert torch.equal(got_in.float().cpu(), want_in), (name, int(t))
            pred = torch.randn(2, 5, 16, 6, 10, generator=g)
            if pred_bf16:
                pred = bf(pred)
            u, c = pred.float().chunk(2)
            v = u + 6.0 * (c - u)
            if s.ancestral if hasattr(s, "ancestral") else False:
                nz = torch.randn(x.shape, generator=gen_ref, device="cuda", dtype=torch.float32).cpu()
                ref = p.R(s.step(p, v, t, x, noise=nz))
            else:
                ref = p.R(s.step(p, v, t, x))
            d = dev(pred)
            got = ps.fused_cfg_step(d[:1], d[1:], xd, 6.0, t, generator=gen)
            assert got.dtype == torch.bfloat16 and torch.equal(got.float().cpu(), ref), (name, n_steps, i, float((got.float().cpu() - ref).abs().max()))
            x, xd = ref.to(torch.bfloat16), got
    # diffusers-shaped `step` (no guidance) == fused form with guidance 1 and no conditional half
    ps, s = pc(), oc()
    ps.set_timesteps(4), s.set_timesteps(4)
    x = bf(torch.randn(1, 3, 16, 4, 6, generator=g))
    u = torch.randn(1, 3, 16, 4, 6, generator=g)
    t = s.timesteps[0]
    kw = dict(noise=torch.randn(x.shape, generator=torch.Generator(device="cuda").manual_seed(9), device="cuda").cpu()) if name == "Euler A" else {}
    got = ps.step(dev(u), t, dev(x), generator=torch.Generator(device="cuda").manual_seed(9))[0]
    assert torch.equal(got.float().cpu(), p.R(s.step(p, u, t, x, **kw)))
    with pytest.raises(ValueError, match="not on the schedule"):
        ps.fused_cfg_step(dev(u), None, dev(x), 1.0, 998)


def test_cfg_pndm_step_matches_oracle(ops):
    """`tcx_cfg_pndm_step` ("PNDM") through the product scheduler's `fused_cfg_step`: the whole 59-evaluation trajectory of a 50-step
    schedule (12 Runge-Kutta evaluations, 47 multistep updates) with CFG 6 on fixed random model outputs, against the oracle's restated
    PNDMScheduler under the bf16 contract — the kernel applies the library's fp32 operations in their order: bit-equal latents after
    every evaluation.  Out-of-order timesteps are refused (the schedule is stateful)."""
    from trajectorycrafter_amd.scheduler import PNDMScheduler
    g = torch.Generator().manual_seed(4711)
    p = Prec("bf16")
    for n_steps, pred_bf16 in ((50, False), (6, True)):
        ps, s = PNDMScheduler(), dr.PNDMScheduler()
        ps.set_timesteps(n_steps), s.set_timesteps(n_steps)
        x = bf(torch.randn(1, 5, 16, 6, 10, generator=g))
        xd = dev(x)
        for i, t in enumerate(s.timesteps.tolist()):
            pred = torch.randn(2, 5, 16, 6, 10, generator=g)
            if pred_bf16:
                pred = bf(pred)
            u, c = pred.float().chunk(2)
            ref = p.R(s.step(p, u + 6.0 * (c - u), t, x))
            d = dev(pred)
            got = ps.fused_cfg_step(d[:1], d[1:], xd, 6.0, t)
            assert got.dtype == torch.bfloat16 and torch.equal(got.float().cpu(), ref), (n_steps, i, t, float((got.float().cpu() - ref).abs().max()))
            x, xd = ref.to(torch.bfloat16), got
        with pytest.raises(RuntimeError, match="more steps"):
            ps.fused_cfg_step(d[:1], d[1:], xd, 6.0, 19)
    ps = PNDMScheduler()
    ps.set_timesteps(10)
    with pytest.raises(ValueError, match="expects timestep"):
        ps.fused_cfg_step(d[:1], d[1:], xd, 6.0, 899)
    # the diffusers-shaped step (no guidance)
    ps, s = PNDMScheduler(), dr.PNDMScheduler()
    ps.set_timesteps(8), s.set_timesteps(8)
    u = torch.randn(1, 5, 16, 6, 10, generator=g)
    assert torch.equal(ps.step(dev(u), 999, xd)[0].float().cpu(), p.R(s.step(p, u, 999, xd.float().cpu())))


# ----------------------------------------------------------------------------- VAE kernels
def to_cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def from_cl(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


def w_cl(w):
    return w.permute(0, 2, 3, 4, 1).contiguous()


@pytest.mark.parametrize("Cin,Cout,T,H,W", [(16, 32, 3, 6, 10), (32, 64, 2, 9, 7), (64, 3, 2, 8, 8), (128, 256, 1, 12, 12)])
def test_causal_conv3d_chunks_with_cache(ops, Cin, Cout, T, H, W):
    g = torch.Generator().manual_seed(Cin + T)
    p = Prec("bf16")
    w = bf(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin))
    b = bf(torch.randn(Cout, generator=g) * 0.1)
    sd = {"c.conv.weight": w.float(), "c.conv.bias": b.float()}
    x1, x2 = bf(torch.randn(1, Cin, T, H, W, generator=g)), bf(torch.randn(1, Cin, 2, H, W, generator=g))
    cache = {}
    r1 = ovae.causal_conv3d(p, sd, "c.", x1.float(), cache)
    r2 = ovae.causal_conv3d(p, sd, "c.", x2.float(), cache)
    dw, db = dev(w_cl(w)), dev(b)
    d1, d2 = dev(to_cl(x1)), dev(to_cl(x2))
    y1 = ops.conv3d_cl(d1, dw, db)                                         # first chunk: replicate first frame
    c = torch.cat([d1[:, :1], d1[:, :1], d1], 1)[:, -2:].contiguous()
    y2 = ops.conv3d_cl(d2, dw, db, cache=c)
    assert_bf16_close(from_cl(y1), r1, atol=2e-3)
    assert_bf16_close(from_cl(y2), r2, atol=2e-3)


def test_conv_pointwise_residual_and_linear(ops):
    g = torch.Generator().manual_seed(11)
    Cin, Cout = 64, 96
    x = bf(torch.randn(1, 2, 5, 7, Cin, generator=g))
    w = bf(torch.randn(Cout, Cin, generator=g) / 8)
    b = bf(torch.randn(Cout, generator=g))
    res = bf(torch.randn(1, 2, 5, 7, Cout, generator=g))
    ref = F.linear(x.float(), w.float(), b.float()) + res.float()
    y = ops.conv3d_cl(dev(x), dev(w.reshape(Cout, 1, 1, 1, Cin)), dev(b), res=dev(res))
    assert_bf16_close(y, ref, atol=2e-3)
    xl = bf(torch.randn(3, 50, 200, generator=g))          # K = 200 (not a multiple of 32), ragged M
    wl = bf(torch.randn(72, 200, generator=g) / 14)
    assert_bf16_close(ops.conv3d_cl(dev(xl).reshape(1, 1, 1, -1, xl.shape[-1]), dev(wl).reshape(wl.shape[0], 1, 1, 1, wl.shape[1]), None).reshape(3, 50, wl.shape[0]),
                      F.linear(xl.float(), wl.float()), atol=2e-3)       # 1x1x1 convolution == plain GEMM


@pytest.mark.parametrize("T,compress", [(3, True), (2, True), (3, False), (1, True)])
def test_upsample_fused_conv2d(ops, T, compress):
    g = torch.Generator().manual_seed(T)
    C, H, W = 32, 5, 6
    p = Prec("bf16")
    x = bf(torch.randn(1, C, T, H, W, generator=g))
    w = bf(torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C))
    b = bf(torch.randn(C, generator=g) * 0.1)
    ref = dr.upsample3d(p, {"conv.weight": w.float(), "conv.bias": b.float()}, "", x.float(), compress)
    from trajectorycrafter_amd.models.autoencoder_magvit import upsample_t_map
    tm = upsample_t_map(T, compress)
    y = ops.conv3d_cl(dev(to_cl(x)), dev(w.permute(0, 2, 3, 1).reshape(C, 1, 3, 3, C).contiguous()), dev(b), ups=1,
                      t_map=torch.tensor(tm, dtype=torch.int32, device="cuda"))
    assert_bf16_close(from_cl(y), ref, atol=2e-3)


@pytest.mark.parametrize("C,G,T,H,W,zT", [(32, 8, 5, 8, 12, 3), (64, 32, 4, 8, 8, 2), (128, 32, 1, 6, 6, 1)])
def test_groupnorm_spatialnorm_silu(ops, C, G, T, H, W, zT):
    g = torch.Generator().manual_seed(C + T)
    p = Prec("bf16")
    f = bf(torch.randn(1, C, T, H, W, generator=g) * 2 + 3)       # large mean: stresses the one-pass variance
    zq = bf(torch.randn(1, 16, zT, H // 2, W // 2, generator=g))
    sd = {"n.norm_layer.weight": bf(1 + 0.1 * torch.randn(C, generator=g)).float(),
          "n.norm_layer.bias": bf(0.1 * torch.randn(C, generator=g)).float(),
          "n.conv_y.conv.weight": bf(torch.randn(C, 16, 1, 1, 1, generator=g) / 4).float(),
          "n.conv_y.conv.bias": bf(1 + 0.1 * torch.randn(C, generator=g)).float(),
          "n.conv_b.conv.weight": bf(torch.randn(C, 16, 1, 1, 1, generator=g) / 4).float(),
          "n.conv_b.conv.bias": bf(0.1 * torch.randn(C, generator=g)).float()}
    ref = ovae.spatial_norm3d(p, sd, "n.", f.float(), zq.float(), G, {})
    from trajectorycrafter_amd.models.autoencoder_magvit import zq_t_map
    x = dev(to_cl(f))
    zcl = dev(to_cl(zq))
    ytab = ops.conv3d_cl(zcl, dev(bf(sd["n.conv_y.conv.weight"]).reshape(C, 1, 1, 1, 16)), dev(bf(sd["n.conv_y.conv.bias"])))
    btab = ops.conv3d_cl(zcl, dev(bf(sd["n.conv_b.conv.weight"]).reshape(C, 1, 1, 1, 16)), dev(bf(sd["n.conv_b.conv.bias"])))
    stats = ops.groupnorm_stats(x, G, 1e-6)
    ref_mean = f.float().reshape(1, G, -1).mean(-1)
    torch.testing.assert_close(stats[..., 0].cpu(), ref_mean, rtol=1e-5, atol=1e-5)
    ref_rstd = (f.float().reshape(1, G, -1).var(-1, unbiased=False) + 1e-6).rsqrt()
    torch.testing.assert_close(stats[..., 1].cpu(), ref_rstd, rtol=1e-4, atol=1e-5)
    tm = torch.tensor(zq_t_map(T, zT), dtype=torch.int32, device="cuda")
    y = ops.groupnorm_apply(x, stats, dev(bf(sd["n.norm_layer.weight"])), dev(bf(sd["n.norm_layer.bias"])), G, ytab, btab, tm)
    assert_bf16_close(from_cl(y), ref, atol=3e-3, ulps=2.0)       # Y/Bt tables are bf16-rounded (reference per-op rounding)
    # plain GroupNorm + SiLU (encoder side)
    y = ops.groupnorm_apply(x, stats, dev(bf(sd["n.norm_layer.weight"])), dev(bf(sd["n.norm_layer.bias"])), G)
    gsd = {"g.weight": sd["n.norm_layer.weight"], "g.bias": sd["n.norm_layer.bias"]}
    assert_bf16_close(from_cl(y), ovae.group_norm_silu(p, gsd, "g.", f.float(), G, 1e-6), atol=2e-3)


def test_layout_kernels(ops):
    g = torch.Generator().manual_seed(6)
    x = bf(torch.randn(2, 16, 3, 5, 7, generator=g))
    got = ops.ncthw_to_cl(dev(x), 1 / 1.15258426)
    assert_bf16_close(got, to_cl(x.float() / 1.15258426), atol=1e-6)
    cl = bf(torch.randn(1, 4, 6, 10, 3, generator=g) * 2)
    out = torch.zeros(1, 3, 9, 6, 10, device="cuda")
    ops.cl_to_frames(dev(cl), out, 5)
    ref = (from_cl(cl).float() / 2 + 0.5).to(BF).float().clamp(0, 1)
    assert torch.equal(out[:, :, 5:].cpu(), ref)
    assert float(out[:, :, :5].abs().max()) == 0


@pytest.mark.parametrize("T,compress", [(5, True), (4, True), (3, False), (1, True)])
def test_downsample_conv_and_avgpool(ops, T, compress):
    """diffusers CogVideoXDownsample3D: temporal avg-pool + stride-2 conv on the (0,1,0,1)-padded frame."""
    g = torch.Generator().manual_seed(T + 40)
    C, H, W = 32, 10, 12
    p = Prec("bf16")
    x = bf(torch.randn(1, C, T, H, W, generator=g))
    w = bf(torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C))
    b = bf(torch.randn(C, generator=g) * 0.1)
    ref = dr.downsample3d(p, {"conv.weight": w.float(), "conv.bias": b.float()}, "", x.float(), compress)
    xcl = dev(to_cl(x))
    if compress and T > 1:
        xcl = ops.avgpool_t(xcl)
    y = ops.conv3d_cl(xcl, dev(w.permute(0, 2, 3, 1).reshape(C, 1, 3, 3, C).contiguous()), dev(b), stride=2, pad=(0, 0),
                      out_hw=(H // 2, W // 2))
    assert_bf16_close(from_cl(y), ref, atol=2e-3)
