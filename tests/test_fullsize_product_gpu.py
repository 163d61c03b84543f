"""Full-size (BASELINE configs[1] / configs[2]: 49 f, 480x720, B = 2 -> S = 17 776 tokens) parity of WHAT SHIPS.  GPU only.

The product does not call the generic attention kernel that tests/test_fullsize_gpu.py samples: it calls
`qk_layernorm_rope(.., q_scale = dh^-1/2 log2 e, want_k_sqmax=True)` -> `attn_fwd(.., log2_scores=True, k_sqmax=..)`
(crosstransformer3d.py Attention.forward), i.e. the bound-centred kernel plus the exact-tracking kernel on the complement
workgroups, and for the cross-attention `scale_bf16` / `scale_sqmax` -> the same pair at D = 128.  Here those exact call chains
run at full size and are compared with

  * fp32 torch arithmetic on sampled rows (attention: the other side of the comparison needs all 17 776 keys but only
    the sampled queries), including a workgroup forced over the M >= 60 predicate so that the complement launch computes, and
  * the ORACLE ITSELF at full size for one whole CogVideoXBlock, one PerceiverCrossAttention and a 2-block
    CrossTransformer3DModel forward: the oracle is device-agnostic torch code (pinned on the CPU against the reference
    fixtures, tests/test_oracle_golden.py); given device tensors the same functions run in fp32 on the GPU through torch, which
    makes an every-element comparison at S = 17 776 take seconds instead of the ~10 minutes the CPU needs per block.
    torch here is the checker's arithmetic engine, never the product's.

Tolerances are stated per test.  Deep chains use tests/test_models_gpu._check_deep (accuracy against the fp32 result must
match the oracle's bf16 rounding contract), single kernels use bf16-ulp bounds.
"""
import argparse
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import diffusers_restated as dr
from oracle import transformer as otr
from oracle.pipeline import prepare_rotary
from oracle.prec import Prec
from tests.test_models_gpu import _check_deep

BF = torch.bfloat16
B, S, H, D, TEXT = 2, 17776, 48, 64, 226
SV, SR = S - TEXT, 4050
LOG2E = 1.4426950408889634


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.backends.cuda.matmul.allow_tf32 = False             # the checker's matmuls are true fp32
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops(gpu):
    from trajectorycrafter_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def rotary(gpu):
    cos, sin = prepare_rotary(480, 720, 13, 2, 64)
    return cos.to(gpu), sin.to(gpu)


def _softmax2_rows(q_rows, k, v):
    """fp32 base-2 attention for sampled query rows: q_rows [R,D], k / v [Sk,D] -> [R,D]."""
    s = q_rows.float() @ k.float().T
    e = torch.exp2(s - s.amax(-1, keepdim=True))
    return (e @ v.float()) / e.sum(-1, keepdim=True)


def test_self_attention_shipped_path_fullsize(ops, gpu, rotary):
    """Attention.forward's chain at [2, 17776, 48, 64] on the fused-QKV layout, vs fp32 on sampled rows.
    Tolerance: outputs average 17 776 values of ~N(0,1): |o| ~ 1e-2; P is rounded to bf16 once (relative 2^-9 per
    probability, averaging out), the output once: max err < 2e-3 and mean err < 2.5e-4 (the same bounds the generic kernel
    meets in test_fullsize_gpu.py).  The LN + RoPE stage is checked on the same rows within one bf16 ulp."""
    cos, sin = rotary
    g = torch.Generator(device=gpu).manual_seed(0)
    qkv = torch.randn(B, S, 3 * H * D, device=gpu, dtype=BF, generator=g)
    raw = qkv.clone()
    q, k, v = (t.view(B, S, H, D) for t in qkv.chunk(3, -1))
    gq, bq, gk, bk = (torch.randn(D, device=gpu, dtype=BF, generator=g) * s + o for s, o in ((0.1, 1.0), (0.05, 0.0), (0.1, 1.0), (0.05, 0.0)))
    qs = D ** -0.5 * LOG2E
    ksq = ops.qk_layernorm_rope(q, k, gq, bq, gk, bk, cos, sin, TEXT, 1e-6, q_scale=qs, want_k_sqmax=True)
    rows = torch.tensor([0, 1, 225, 226, 227, 4097, 8888, 17000, 17519, 17520, 17775], device=gpu)   # 17520 = start of the ragged last q-block
    heads = [0, 17, 47]

    # (1) LN + RoPE (+ q scale) on sampled rows, fp32 reference from the raw projections (oracle: dr.cogvideox_attention)
    rq, rk, _ = (t.view(B, S, H, D) for t in raw.chunk(3, -1))
    for src, got, gam, bet, scl in ((rq, q, gq, bq, qs), (rk, k, gk, bk, 1.0)):
        x = src[:, rows].float()                                                     # [B,R,H,D]
        n = F.layer_norm(x, (D,), gam.float(), bet.float(), 1e-6)
        vid = rows >= TEXT
        pos = (rows - TEXT).clamp(min=0)
        rot = dr.apply_rotary_emb(n.permute(0, 2, 1, 3), cos[pos], sin[pos]).permute(0, 2, 1, 3)
        ref = torch.where(vid[None, :, None, None], rot, n) * scl
        err = (got[:, rows].float() - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -7 + 1e-3).all()), float(err.max())
    # k_sqmax is what it says: max over tokens of |k|^2 per (batch, head), from the rounded k the attention reads
    want_ksq = (k.float() ** 2).sum(-1).amax(1)
    assert bool(((ksq - want_ksq).abs() <= 1e-3 * want_ksq).all())

    # (2) every workgroup safe (M = |q| max|k| ~ 12 < 60): the bound-centred kernel computes everything
    o = ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq)
    assert torch.isfinite(o.float()).all()
    emax = emean = 0.0
    for h in heads:
        for b in range(B):
            ref = _softmax2_rows(q[b, rows, h], k[b, :, h], v[b, :, h])
            err = (o[b, rows, h].float() - ref).abs()
            emax, emean = max(emax, float(err.max())), max(emean, float(err.mean()))
    print(f"self-attention shipped path, sampled rows: max err {emax:.3e}, worst mean err {emean:.3e}")
    assert emax < 2e-3 and emean < 2.5e-4

    # (2b) TCX_ATTN_BOUND_PROVEN (what the model passes when the LayerNorm parameters prove M < 60): no per-workgroup test, no
    #      complement launch -> the same bits
    assert torch.equal(ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq, bound_proven=True), o)

    # (3) one workgroup forced over the predicate: rows 1024..1030 of (b=1, h=5) scaled x6 -> M ~ 70 >= 60 for q-block 4
    #     (rows 1024..1279).  That workgroup must be computed by the complement launch = the exact-tracking kernel:
    #     bit-identical to the k_sqmax=None run there, while safe workgroups (different centring) are not all identical.
    q[1, 1024:1031, 5] = (q[1, 1024:1031, 5].float() * 6.0).to(BF)
    ob = ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq)
    oe = ops.attn_fwd(q, k, v, 1.0, log2_scores=True)                               # exact-tracking FAST loop everywhere
    assert torch.equal(ob[1, 1024:1280, 5], oe[1, 1024:1280, 5])
    assert not torch.equal(ob[1, :, 4], oe[1, :, 4])                                # the bound-centred loop did run elsewhere
    r2 = torch.tensor([1023, 1024, 1027, 1030, 1031, 1279, 1280], device=gpu)
    ref = _softmax2_rows(q[1, r2, 5], k[1, :, 5], v[1, :, 5])
    err = (ob[1, r2, 5].float() - ref).abs()
    # the x6 rows are peaked (a few keys dominate): error scales with |o| (up to ~1): one bf16 ulp of the value + 2e-3
    assert bool((err <= ref.abs() * 2.0 ** -7 + 2e-3).all()), float(err.max())
    # untouched heads / batches are unchanged by the forced workgroup
    assert torch.equal(ob[0], o[0]) and torch.equal(ob[1, :, :5], o[1, :, :5]) and torch.equal(ob[1, :, 6:], o[1, :, 6:])


def test_self_attention_tail_split_fullsize(ops, gpu):
    """The bound-centred launch cuts the workgroups of its last, partly filled round (6720 mod 256 = 64 of them at this
    shape) into 4 key ranges and adds the parts (tcx_attn_fwd_ws, include/tcx_hip.h).  Against the single-pass launch
    (split_tail=False): bit-identical on every workgroup that is not split; on the split ones only the fp32 summation order
    differs (the parts share the exponent origin): |diff| <= 1 bf16 ulp of the value + 1e-4, and the same fp32-reference bound as
    the unsplit kernel; a split workgroup that fails the M < 60 predicate is left to the exact kernel, bit for bit."""
    lib = __import__("trajectorycrafter_amd._lib", fromlist=["load"]).load()
    need = int(lib.tcx_attn_fwd_workspace_bytes(B, H, S, S, D, 1, 1, 0))
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    nwg = 70 * B * H
    if nwg % cus == 0 or 2 * (nwg % cus) > cus:
        assert need == 0
        pytest.skip(f"no tail to split on a {cus}-CU device")
    tail, split = nwg % cus, min(cus // (nwg % cus), 8)
    assert need == (tail * split * 256 * (D + 1) + tail * split) * 4
    g = torch.Generator(device=gpu).manual_seed(5)
    q = torch.randn(B, S, H, D, device=gpu, dtype=BF, generator=g) * (D ** -0.5 * LOG2E)
    k = torch.randn(B, S, H, D, device=gpu, dtype=BF, generator=g)
    v = torch.randn(B, S, H, D, device=gpu, dtype=BF, generator=g)
    ksq = (k.float() ** 2).sum(-1).amax(1).contiguous()
    a = ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq)                          # product call: split
    b_ = ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq, split_tail=False)
    same = (a == b_).flatten(3).all(-1)                                                   # [B,S,H] rows that agree bit for bit
    blocks = same.view(B, -1, H)[:, : (S // 256) * 256].reshape(B, S // 256, 256, H).all(2)   # per (b, q-block, h), full blocks
    n_diff = int((~blocks).sum()) + int(not bool(same[:, (S // 256) * 256:].all(1).all()))
    assert 0 < n_diff <= tail, (n_diff, tail)                                              # only split workgroups may differ
    d = (a.float() - b_.float()).abs()
    assert bool((d <= b_.float().abs() * 2.0 ** -7 + 1e-4).all()), float(d.max())
    bb, qq, hh = [int(x[0]) for x in torch.nonzero(~blocks, as_tuple=True)]                # one split workgroup: check it vs fp32
    rows = torch.arange(qq * 256, qq * 256 + 256, 37, device=gpu)
    ref = _softmax2_rows(q[bb, rows, hh], k[bb, :, hh], v[bb, :, hh])
    err = (a[bb, rows, hh].float() - ref).abs()
    assert float(err.max()) < 2e-3 and float(err.mean()) < 2.5e-4
    # force that workgroup over the predicate: every part returns, the combine kernel skips it, the exact kernel computes it
    q[bb, qq * 256 + 3: qq * 256 + 9, hh] = (q[bb, qq * 256 + 3: qq * 256 + 9, hh].float() * 6.0).to(BF)
    c = ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq)
    e = ops.attn_fwd(q, k, v, 1.0, log2_scores=True)
    assert torch.equal(c[bb, qq * 256: qq * 256 + 256, hh], e[bb, qq * 256: qq * 256 + 256, hh])
    assert torch.isfinite(c.float()).all()
    # repeatable (parts are added in a fixed order)
    assert torch.equal(c, ops.attn_fwd(q, k, v, 1.0, log2_scores=True, k_sqmax=ksq))


def test_cross_attention_shipped_path_fullsize(ops, gpu):
    """PerceiverCrossAttention.forward's chain at q [2,17550,16,128] x k/v [2,4050,16,128] (k, v strided halves of one
    to_kv output as in the model) vs fp32 on sampled rows.  Tolerance as above (|o| ~ 1/sqrt(4050) ~ 1.6e-2)."""
    g = torch.Generator(device=gpu).manual_seed(2)
    Hc, Dc = 16, 128
    q = torch.randn(B, SV, Hc * Dc, device=gpu, dtype=BF, generator=g)
    kv = torch.randn(B, SR, 2 * Hc * Dc, device=gpu, dtype=BF, generator=g)
    k_raw, v = kv.chunk(2, -1)
    s = Dc ** -0.25
    qq = ops.scale_bf16(q, s * LOG2E)
    kk, ksq = ops.scale_sqmax(k_raw, s, Hc, Dc)
    assert torch.equal(kk, (k_raw.float() * s).to(BF))                               # single rounding of k * s
    want = (kk.view(B, SR, Hc, Dc).float() ** 2).sum(-1).amax(1)
    assert bool(((ksq - want).abs() <= 1e-3 * want).all())
    q4, k4, v4 = qq.view(B, SV, Hc, Dc), kk.view(B, SR, Hc, Dc), v.view(B, SR, Hc, Dc)
    o = ops.attn_fwd(q4, k4, v4, 1.0, log2_scores=True, k_sqmax=ksq)
    rows = torch.tensor([0, 255, 256, 9000, 17407, 17408, 17549], device=gpu)        # 17408 = start of the ragged last q-block
    emax = emean = 0.0
    for h in (0, 7, 15):
        for b in range(B):
            ref = _softmax2_rows(q4[b, rows, h], k4[b, :, h], v4[b, :, h])
            err = (o[b, rows, h].float() - ref).abs()
            emax, emean = max(emax, float(err.max())), max(emean, float(err.mean()))
    print(f"cross-attention shipped path, sampled rows: max err {emax:.3e}, worst mean err {emean:.3e}")
    assert emax < 4e-3 and emean < 5e-4
    # forced-unsafe workgroup (M >= 60 for q-block 2 of (b=0, h=3)): complement launch, bit-identical to the exact loop there
    q4[0, 512:520, 3] = (q4[0, 512:520, 3].float() * 5.0).to(BF)
    ob = ops.attn_fwd(q4, k4, v4, 1.0, log2_scores=True, k_sqmax=ksq)
    oe = ops.attn_fwd(q4, k4, v4, 1.0, log2_scores=True)
    assert torch.equal(ob[0, 512:768, 3], oe[0, 512:768, 3]) and not torch.equal(ob[0, :, 2], oe[0, :, 2])
    r2 = torch.tensor([511, 512, 515, 519, 520, 767, 768], device=gpu)
    ref = _softmax2_rows(q4[0, r2, 3], k4[0, :, 3], v4[0, :, 3])
    err = (ob[0, r2, 3].float() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -7 + 4e-3).all()), float(err.max())


def _block_sd(gpu, prefixes, seed=0):
    from trajectorycrafter_amd import init_weights as iw
    cfg = dict(otr.DEFAULT_CONFIG, **dict(iw.TRANSFORMER_5B, num_layers=2))
    shapes = {k: v for k, v in iw.transformer_param_shapes(cfg).items() if k.startswith(prefixes)}
    sd = iw.random_state_dict(shapes, seed=seed, dtype=torch.float32, device=gpu)
    return {k: v.to(BF).float() for k, v in sd.items()}                              # bf16-representable weights


def test_block_fullsize_vs_oracle(gpu, rotary):
    """configs[1]: one CogVideoXBlock.forward (reference :224-266) at hidden [2,17550,3072], encoder [2,226,3072], temb
    [2,512] ~ N(0,1), every element against the oracle's block (bf16 contract and fp32) evaluated at full size."""
    from trajectorycrafter_amd.models.crosstransformer3d import CogVideoXBlock
    pre = "transformer_blocks.0."
    sd = _block_sd(gpu, (pre,))
    blk = CogVideoXBlock(dim=H * D, num_attention_heads=H, attention_head_dim=D, time_embed_dim=512, attention_bias=True)
    blk.load_state_dict({k[len(pre):]: v for k, v in sd.items()}, strict=True)
    blk = blk.to(gpu, BF).eval()
    g = torch.Generator(device=gpu).manual_seed(1)
    hidden = torch.randn(B, SV, H * D, device=gpu, dtype=BF, generator=g)
    enc = torch.randn(B, TEXT, H * D, device=gpu, dtype=BF, generator=g)
    temb = torch.randn(B, 512, device=gpu, dtype=BF, generator=g)
    with torch.no_grad():
        h_hip, e_hip = blk(hidden, enc, temb, image_rotary_emb=rotary)
        h_con, e_con = otr.cogvideox_block(Prec("bf16"), sd, pre, hidden.float(), enc.float(), temb.float(), rotary, H, 1e-5)
        h_ex, e_ex = otr.cogvideox_block(Prec("fp32"), sd, pre, hidden.float(), enc.float(), temb.float(), rotary, H, 1e-5)
    _check_deep(h_hip, h_con, h_ex, "full-size CogVideoXBlock, video rows")
    _check_deep(e_hip, e_con, e_ex, "full-size CogVideoXBlock, text rows")
    # element-wise: both sides round at the same tensors, so they differ by accumulation order and the rare one-ulp flip of
    # an intermediate: 99.9 % within 2 bf16 ulps (+ atol for the near-zero values), nothing beyond 16 ulps
    for got, con in ((h_hip, h_con), (e_hip, e_con)):
        err = (got.float() - con).abs()
        ulp = con.abs() * 2.0 ** -7
        assert float((err > 2 * ulp + 2e-2).float().mean()) < 1e-3, float((err > 2 * ulp + 2e-2).float().mean())
        assert bool((err <= 16 * ulp + 0.25).all()), float(err.max())


def test_perceiver_cross_attention_fullsize_vs_oracle(gpu):
    """configs[1]: one PerceiverCrossAttention.forward (reference :376-398) at x [2,4050,3072], latents [2,17550,3072]."""
    from trajectorycrafter_amd.models.crosstransformer3d import PerceiverCrossAttention
    pre = "perceiver_cross_attention.0."
    sd = _block_sd(gpu, (pre,), seed=3)
    pca = PerceiverCrossAttention(dim=H * D, dim_head=128, heads=16, kv_dim=None)
    pca.load_state_dict({k[len(pre):]: v for k, v in sd.items()}, strict=True)
    pca = pca.to(gpu, BF).eval()
    g = torch.Generator(device=gpu).manual_seed(4)
    x = torch.randn(B, SR, H * D, device=gpu, dtype=BF, generator=g)
    lat = torch.randn(B, SV, H * D, device=gpu, dtype=BF, generator=g)
    with torch.no_grad():
        got = pca(x, lat)
        con = Prec("bf16").R(otr.perceiver_cross_attention(Prec("bf16"), sd, pre, x.float(), lat.float(), 16, 128))
        ex = otr.perceiver_cross_attention(Prec("fp32"), sd, pre, x.float(), lat.float(), 16, 128)
        _check_deep(got, con, ex, "full-size PerceiverCrossAttention")
        # the model's form: latents += to_out(...) in the GEMM epilogue (:833-837)
        lat2 = lat.clone()
        pca(x, lat2, add_to_latents=True)
        con2 = Prec("bf16").R(lat.float() + otr.perceiver_cross_attention(Prec("bf16"), sd, pre, x.float(), lat.float(), 16, 128))
        _check_deep(lat2, con2, lat.float() + ex, "full-size PerceiverCrossAttention + residual")


def test_two_block_transformer_forward_fullsize_vs_oracle(gpu, rotary):
    """CrossTransformer3DModel.forward (reference :711-871) at the configs[2] input shapes (hidden [2,13,16,60,90], inpaint
    [2,13,17,60,90], cross [2,3,16,60,90], text [2,226,4096]) with num_layers = 2 (2 blocks + 1 cross layer + embeddings +
    norm_out / proj_out / unpatchify: every module of the 42-layer model, at full token count) against the oracle."""
    from trajectorycrafter_amd import init_weights as iw
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    cfg = dict(iw.TRANSFORMER_5B, num_layers=2)
    sd32 = iw.random_state_dict(iw.transformer_param_shapes(dict(otr.DEFAULT_CONFIG, **cfg)), seed=0, dtype=torch.float32, device=gpu)
    sd32 = {k: v.to(BF).float() for k, v in sd32.items()}
    with torch.device("meta"):
        model = CrossTransformer3DModel(**cfg)
    model.load_state_dict({k: v.to(BF) for k, v in sd32.items()}, strict=True, assign=True)
    model.eval()
    g = torch.Generator(device=gpu).manual_seed(7)
    rn = lambda *s: torch.randn(*s, device=gpu, dtype=BF, generator=g)
    hs, txt, inp, cross = rn(2, 13, 16, 60, 90), rn(2, 226, 4096), rn(2, 13, 17, 60, 90), rn(2, 3, 16, 60, 90)
    ts = torch.tensor([999, 999], device=gpu)
    with torch.no_grad():
        got = model(hs, txt, ts, inpaint_latents=inp, cross_latents=cross, image_rotary_emb=rotary, return_dict=False)[0]
        con = otr.transformer_forward(sd32, cfg, hs.float(), txt.float(), ts, inp.float(), cross.float(), rotary, prec="bf16")
        ex = otr.transformer_forward(sd32, cfg, hs.float(), txt.float(), ts, inp.float(), cross.float(), rotary, prec="fp32")
    assert got.shape == (2, 13, 16, 60, 90) and got.dtype == BF
    _check_deep(got, con, ex, "full-size 2-block CrossTransformer3DModel.forward")


def test_two_block_2b_width_sincos_forward_fullsize_vs_oracle(gpu):
    """The NON-rotary model family (CogVideoX-Fun-2B geometry: 30 heads x 64 = 1920 wide, sincos position table added to the
    joint sequence, reference :752-784; no RoPE in the attention processor) at the configs[2] input shapes with num_layers = 2:
    GEMMs with N = 1920 / 5760 / 7680, LayerNorm rows of 1920, 30-head attention, `tcx_gated_residual` for the table."""
    from trajectorycrafter_amd import init_weights as iw
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    cfg = dict(iw.TRANSFORMER_5B, num_attention_heads=30, num_layers=2, use_rotary_positional_embeddings=False)
    sd32 = iw.random_state_dict(iw.transformer_param_shapes(dict(otr.DEFAULT_CONFIG, **cfg)), seed=5, dtype=torch.float32, device=gpu)
    sd32 = {k: v.to(BF).float() for k, v in sd32.items()}
    model = CrossTransformer3DModel(**cfg)
    model.load_state_dict({k: v.cpu() for k, v in sd32.items()}, strict=True)
    model = model.to(gpu, BF).eval()
    assert model.pos_embedding.shape == (1, 226 + 17550, 1920) and model.pos_embedding.dtype == BF
    g = torch.Generator(device=gpu).manual_seed(9)
    rn = lambda *s: torch.randn(*s, device=gpu, dtype=BF, generator=g)
    hs, txt, inp, cross = rn(2, 13, 16, 60, 90), rn(2, 226, 4096), rn(2, 13, 17, 60, 90), rn(2, 3, 16, 60, 90)
    ts = torch.tensor([999, 999], device=gpu)
    with torch.no_grad():
        got = model(hs, txt, ts, inpaint_latents=inp, cross_latents=cross, image_rotary_emb=None, return_dict=False)[0]
        con = otr.transformer_forward(sd32, cfg, hs.float(), txt.float(), ts, inp.float(), cross.float(), None, prec="bf16")
        ex = otr.transformer_forward(sd32, cfg, hs.float(), txt.float(), ts, inp.float(), cross.float(), None, prec="fp32")
    assert got.shape == (2, 13, 16, 60, 90) and got.dtype == BF
    _check_deep(got, con, ex, "full-size 2-block non-rotary (2B width) CrossTransformer3DModel.forward")


def test_reference_default_resolution_384x672_full_model(gpu):
    """The reference's own default sample size (demo.py / inference.py `--sample_size 384 672`; fractional RoPE grid positions,
    SURVEY §8a row a8; S = 13 330 tokens, 53 q-blocks, 48 x 84 latents): the 42-layer model, 2 steps + decode, finite, in [0, 1],
    bit-repeatable; and the 2-block forward at that size against the oracle."""
    import bench
    from trajectorycrafter_amd import init_weights as iw
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    args = argparse.Namespace(layers=42, frames=49, height=384, width=672)
    pipe = bench.build_models(args, gpu)
    inp = bench.make_inputs(args, gpu, seed=44)
    kw = dict(prompt=None, height=384, width=672, num_frames=49, num_inference_steps=2, guidance_scale=6.0, output_type="pt", **inp)
    a = pipe(**kw).videos
    assert a.shape == (1, 3, 49, 384, 672) and torch.isfinite(a).all() and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert torch.equal(a, pipe(**kw).videos)
    del pipe, a
    cfg = dict(iw.TRANSFORMER_5B, num_layers=2)
    sd32 = iw.random_state_dict(iw.transformer_param_shapes(dict(otr.DEFAULT_CONFIG, **cfg)), seed=0, dtype=torch.float32, device=gpu)
    sd32 = {k: v.to(BF).float() for k, v in sd32.items()}
    with torch.device("meta"):
        model = CrossTransformer3DModel(**cfg)
    model.load_state_dict({k: v.to(BF) for k, v in sd32.items()}, strict=True, assign=True)
    model.eval()
    cos, sin = prepare_rotary(384, 672, 13, 2, 64)
    rot = (cos.to(gpu), sin.to(gpu))
    g = torch.Generator(device=gpu).manual_seed(8)
    rn = lambda *s: torch.randn(*s, device=gpu, dtype=BF, generator=g)
    hs, txt, inp2, cross = rn(2, 13, 16, 48, 84), rn(2, 226, 4096), rn(2, 13, 17, 48, 84), rn(2, 3, 16, 48, 84)
    ts = torch.tensor([499, 499], device=gpu)
    with torch.no_grad():
        got = model(hs, txt, ts, inpaint_latents=inp2, cross_latents=cross, image_rotary_emb=rot, return_dict=False)[0]
        con = otr.transformer_forward(sd32, cfg, hs.float(), txt.float(), ts, inp2.float(), cross.float(), rot, prec="bf16")
        ex = otr.transformer_forward(sd32, cfg, hs.float(), txt.float(), ts, inp2.float(), cross.float(), rot, prec="fp32")
    _check_deep(got, con, ex, "384x672 2-block CrossTransformer3DModel.forward")


@pytest.mark.parametrize("frames,height,width,batch,ref_frames", [(25, 320, 576, 2, 3), (17, 512, 512, 1, 2), (5, 272, 400, 2, 1), (49, 256, 384, 3, 3)])
def test_two_block_forward_other_resolutions_vs_oracle(gpu, frames, height, width, batch, ref_frames):
    """Sizes the reference accepts besides its two defaults (any H, W divisible by 16, <= 49 frames): ragged last q-blocks and key
    tiles in both attentions (S = 5266 / 5346 / 1076 / 5218 + 226 text tokens), ragged GEMM M tiles, odd latent grids (17 x 25
    patches), batch 1 / 2 / 3, fractional RoPE grids — the full-width 2-block model against the oracle, every element."""
    from trajectorycrafter_amd import init_weights as iw
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    cfg = dict(iw.TRANSFORMER_5B, num_layers=2)
    sd32 = iw.random_state_dict(iw.transformer_param_shapes(dict(otr.DEFAULT_CONFIG, **cfg)), seed=0, dtype=torch.float32, device=gpu)
    sd32 = {k: v.to(BF).float() for k, v in sd32.items()}
    with torch.device("meta"):
        model = CrossTransformer3DModel(**cfg)
    model.load_state_dict({k: v.to(BF) for k, v in sd32.items()}, strict=True, assign=True)
    model.eval()
    T, h, w = (frames - 1) // 4 + 1, height // 8, width // 8
    cos, sin = prepare_rotary(height, width, T, 2, 64)
    rot = (cos.to(gpu), sin.to(gpu))
    g = torch.Generator(device=gpu).manual_seed(frames + height)
    rn = lambda *s: torch.randn(*s, device=gpu, dtype=BF, generator=g)
    hs, txt, inp, cross = rn(batch, T, 16, h, w), rn(batch, 226, 4096), rn(batch, T, 17, h, w), rn(batch, ref_frames, 16, h, w)
    ts = torch.full((batch,), 759, device=gpu)
    with torch.no_grad():
        got = model(hs, txt, ts, inpaint_latents=inp, cross_latents=cross, image_rotary_emb=rot, return_dict=False)[0]
        con = otr.transformer_forward(sd32, cfg, hs.float(), txt.float(), ts, inp.float(), cross.float(), rot, prec="bf16")
        ex = otr.transformer_forward(sd32, cfg, hs.float(), txt.float(), ts, inp.float(), cross.float(), rot, prec="fp32")
    assert got.shape == hs.shape
    _check_deep(got, con, ex, f"{frames}f {height}x{width} B={batch} 2-block CrossTransformer3DModel.forward")
    with torch.no_grad():
        assert torch.equal(model(hs, txt, ts, inpaint_latents=inp, cross_latents=cross, image_rotary_emb=rot, return_dict=False)[0], got)


def test_configs2_full_model_two_steps_and_decode(gpu):
    """BASELINE configs[2] smoke at full size: the 42-layer / 6.1 B-parameter model, 2 DDIM steps with CFG + the VAE decode to
    49 frames 480x720 through `TrajCrafter_Pipeline.__call__`: finite, in [0, 1], every frame differs (the decode used all
    latent frames) and the whole call is bit-repeatable (no race in 2 x (42 attention + 200 GEMM) + 700 conv launches)."""
    import bench
    args = argparse.Namespace(layers=42, frames=49, height=480, width=720)
    pipe = bench.build_models(args, gpu)
    inp = bench.make_inputs(args, gpu, seed=43)
    kw = dict(prompt=None, height=480, width=720, num_frames=49, num_inference_steps=2, guidance_scale=6.0, output_type="pt", **inp)
    a = pipe(**kw).videos
    assert a.shape == (1, 3, 49, 480, 720) and a.dtype == torch.float32 and a.is_cuda
    assert torch.isfinite(a).all() and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert float(a.std()) > 1e-3 and float((a[:, :, 1:] - a[:, :, :-1]).abs().amax(dim=(0, 1, 3, 4)).min()) > 0
    b = pipe(**kw).videos
    assert torch.equal(a, b)
    lat = pipe(**dict(kw, output_type="latent")).videos
    assert lat.shape == (1, 13, 16, 60, 90) and lat.dtype == BF and torch.isfinite(lat.float()).all()
    # 2 of 50 steps from pure noise with random weights: the latents are still O(1) noise, not blown up or collapsed
    assert 0.3 < float(lat.float().std()) < 30.0         # (measured 3.7: guidance 6 on random-weight predictions)


def test_oracle_on_device_equals_oracle_on_cpu(gpu):
    """The full-size tests above evaluate the oracle's own functions on device tensors.  Same code, same fp32 arithmetic: at a
    size the CPU finishes at once the two evaluations of a CogVideoXBlock + PerceiverCrossAttention agree to fp32 round-off, in
    both precision modes (the CPU side is the one pinned by the reference fixtures, tests/test_oracle_golden.py)."""
    from trajectorycrafter_amd import init_weights as iw
    cfg = dict(otr.DEFAULT_CONFIG, num_attention_heads=4, num_layers=2, in_channels=33, text_embed_dim=64, time_embed_dim=64,
               use_rotary_positional_embeddings=True, is_train_cross=True, cross_attn_dim_head=128, cross_attn_num_heads=2)
    sd = {k: v.to(BF).float() for k, v in iw.random_state_dict(iw.transformer_param_shapes(cfg), seed=2).items()}
    sdg = {k: v.to(gpu) for k, v in sd.items()}
    g = torch.Generator().manual_seed(3)
    hidden, enc, temb = torch.randn(2, 3 * 4 * 6, 256, generator=g), torch.randn(2, 10, 256, generator=g), torch.randn(2, 64, generator=g)
    ref_tokens = torch.randn(2, 48, 256, generator=g)
    cos, sin = prepare_rotary(64, 96, 3, 2, 64)
    for mode in ("fp32", "bf16"):
        p = Prec(mode)
        hc, ec = otr.cogvideox_block(p, sd, "transformer_blocks.0.", hidden, enc, temb, (cos, sin), 4, 1e-5)
        hg, eg = otr.cogvideox_block(p, sdg, "transformer_blocks.0.", hidden.to(gpu), enc.to(gpu), temb.to(gpu), (cos.to(gpu), sin.to(gpu)), 4, 1e-5)
        cc = otr.perceiver_cross_attention(p, sd, "perceiver_cross_attention.0.", ref_tokens, hc, 2, 128)
        cg = otr.perceiver_cross_attention(p, sdg, "perceiver_cross_attention.0.", ref_tokens.to(gpu), hg, 2, 128)
        for a, b in ((hc, hg), (ec, eg), (cc, cg)):
            d = (a - b.cpu()).abs()
            if mode == "fp32":
                assert float(d.max()) <= 2e-5 * (1 + float(a.abs().max())), float(d.max())
            else:   # a value on a bf16 rounding boundary may flip on fp32 round-off at any of the chain's contract roundings (measured:
                    # 0.5 % of a block's outputs, one ulp each) and the flips feed the next op: never more than one ulp + atol, and a
                    # mean difference far below the rounding step itself
                assert bool((d <= a.abs() * 2.0 ** -7 + 1e-2).all()), float(d.max())
                assert float(d.mean()) <= 2.0 ** -10 * float(a.abs().mean()), (float(d.mean()), float(a.abs().mean()))
