"""The VAE as the PRODUCT dispatches it: `AutoencoderKLCogVideoX()` at the reference's default widths (128 / 256 / 256 / 512,
reference models/autoencoder_magvit.py:991-1024), composed decoder / encoder against the oracle.  GPU only.

tests/test_models_gpu.py checks the composed model on the committed `vae_tiny` fixture, whose channel widths (8/16/16/32) route
every convolution to conv.hip's register-staged kernel.  At the default widths `tcx_conv3d_cl` dispatches

    conv_mfma_kernel<2,4> (Cout >= 256)  |  conv_mfma_kernel<4,2> (Cout == 128)  |  conv_narrow_kernel (decoder conv_out 128 -> 3)
    conv_igemm_kernel (conv_in 16 -> 512 / 8 -> 128, SpatialNorm 1x1x1 tables, encoder conv_out 512 -> 32)

plus the row-structured GroupNorm / SpatialNorm apply kernel with the Bresenham zq column, the folded upsample / stride-2 gathers
and `tcx_avgpool_t` — `test_route_at_default_widths` pins that routing through `tcx_conv3d_route`.  Here that composition runs

  * on a small clip (17 frames 64x96: two decode chunks 3 + 2 latent frames with the conv cache, T == 1, five + four frame
    encode chunks) and
  * at the metric resolution 480x720 on the FIRST TWO chunks of the 49-frame clip (decode: latent frames 0..4 -> 17 frames;
    encode: frames 0..8 -> 3 latent frames) — the chunks that exercise both the first-chunk frame replication and the cache,

against the oracle evaluated in fp32 (the reference's maths) and under the bf16 rounding contract.  The oracle is device-agnostic
torch code pinned on the CPU against the reference's fixtures (tests/test_oracle_golden.py); handed device tensors the same
functions run in fp32 through torch on the GPU (torch is the checker's arithmetic here, never the product's).

Tolerance (every element compared): `_check_deep` — measured against the fp32 result the HIP output must be as accurate as the
oracle's bf16 contract (mean error <= 1.5x, p99.9 <= 2x) and sit as close to the contract as the contract sits to fp32.  Weights
are random (tests/../init_weights.py recipe): no checkpoint exists offline.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae as ovae
from tests.test_models_gpu import _check_deep

BF = torch.bfloat16


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def vae_default(gpu):
    """Default-config model + its state dict (fp32 copies of the bf16 weights, on the device, for the oracle)."""
    from trajectorycrafter_amd.init_weights import random_state_dict, vae_param_shapes
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    vae = AutoencoderKLCogVideoX()
    cfg = dict(vae.config)
    sd = random_state_dict(vae_param_shapes(cfg), seed=11)
    vae.load_state_dict(sd, strict=True)
    vae = vae.to(gpu, BF).eval()
    sdf = {k: v.to(gpu).to(BF).float() for k, v in sd.items()}
    return vae, cfg, sdf


def _tap_conv(x, w, b=None, stride=1, padding=0):
    """N-d convolution as a sum over the kernel taps of channel matmuls in fp32 (`y += x[.., tap window, :] @ w[:, :, tap]^T`):
    the same arithmetic as F.conv2d / F.conv3d up to fp32 summation order.  Only here because MIOpen's fp32 conv3d takes minutes
    per layer at 128 x 9 x 480 x 720; torch matmul is the checker's engine instead (validated against F.conv3d below)."""
    nd = x.dim() - 2
    st = (stride,) * nd if isinstance(stride, int) else tuple(stride)
    pd = (padding,) * nd if isinstance(padding, int) else tuple(padding)
    if any(pd):
        x = torch.nn.functional.pad(x, [v for q in reversed(pd) for v in (q, q)])
    xl = x.movedim(1, -1)                                              # [B, *spatial, C]
    ks = w.shape[2:]
    out_sp = [(xl.shape[1 + i] - ks[i]) // st[i] + 1 for i in range(nd)]
    y = None
    import itertools
    for tap in itertools.product(*[range(k) for k in ks]):
        idx = (slice(None),) + tuple(slice(tap[i], tap[i] + (out_sp[i] - 1) * st[i] + 1, st[i]) for i in range(nd))
        wt = w[(slice(None), slice(None)) + tap]                       # [O, C]
        t = torch.matmul(xl[idx], wt.t())
        y = t if y is None else y.add_(t)
    if b is not None:
        y = y + b
    return y.movedim(-1, 1).contiguous()


@pytest.fixture()
def fast_oracle_convs(monkeypatch, gpu):
    """Route the oracle's F.conv2d / F.conv3d through `_tap_conv` for the duration of a full-size test, after checking it
    against the real F.conv3d / F.conv2d on the device at small sizes (stride, padding, 1x1x1 and 3x3x3 kernels)."""
    import torch.nn.functional as F
    g = torch.Generator(device=gpu).manual_seed(1)
    x3 = torch.randn(2, 24, 5, 14, 18, device=gpu, generator=g)
    for k in ((3, 3, 3), (1, 1, 1)):
        w = torch.randn(40, 24, *k, device=gpu, generator=g) * 0.1
        b = torch.randn(40, device=gpu, generator=g)
        torch.testing.assert_close(_tap_conv(x3, w, b), F.conv3d(x3, w, b), rtol=1e-4, atol=1e-4)
    x2 = torch.randn(3, 24, 15, 19, device=gpu, generator=g)
    w2 = torch.randn(32, 24, 3, 3, device=gpu, generator=g) * 0.1
    b2 = torch.randn(32, device=gpu, generator=g)
    torch.testing.assert_close(_tap_conv(x2, w2, b2, stride=2, padding=0), F.conv2d(x2, w2, b2, stride=2, padding=0), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(_tap_conv(x2, w2, b2, stride=1, padding=1), F.conv2d(x2, w2, b2, stride=1, padding=1), rtol=1e-4, atol=1e-4)
    monkeypatch.setattr(F, "conv3d", lambda x, w, b=None, stride=1, padding=0: _tap_conv(x, w, b, stride, padding))
    monkeypatch.setattr(F, "conv2d", lambda x, w, b=None, stride=1, padding=0: _tap_conv(x, w, b, stride, padding))


def _oracle_decode(sdf, cfg, z):
    with torch.no_grad():
        return (ovae.vae_decode(sdf, cfg, z.float(), prec="bf16").float(), ovae.vae_decode(sdf, cfg, z.float(), prec="fp32").float())


def _oracle_encode(sdf, cfg, x):
    with torch.no_grad():
        con, ex = ovae.vae_encode(sdf, cfg, x.float(), prec="bf16"), ovae.vae_encode(sdf, cfg, x.float(), prec="fp32")
    return con, ex


def test_route_at_default_widths(gpu):
    """Which kernel `tcx_conv3d_cl` launches for the shapes of the default decoder / encoder (csrc/conv.hip dispatch)."""
    from trajectorycrafter_amd import ops
    MFMA_WIDE, MFMA_TALL, NARROW, IGEMM = 1, 2, 3, 4
    r = ops.conv3d_route
    assert r(Cin=512, Cout=512, k=(3, 3, 3)) == MFMA_WIDE                  # mid block / up0
    assert r(Cin=512, Cout=256, k=(3, 3, 3)) == MFMA_WIDE                  # up1 first resnet (+ its 1x1x1 shortcut below)
    assert r(Cin=512, Cout=256, k=(1, 1, 1)) == MFMA_WIDE
    assert r(Cin=256, Cout=256, k=(1, 3, 3), ups=1) == MFMA_WIDE           # upsample conv
    assert r(Cin=256, Cout=128, k=(3, 3, 3)) == MFMA_TALL                  # up3 first resnet: the 512 x 128 tile
    assert r(Cin=128, Cout=128, k=(3, 3, 3)) == MFMA_TALL
    assert r(Cin=128, Cout=128, k=(1, 3, 3), stride=2) == MFMA_TALL        # encoder downsample
    assert r(Cin=128, Cout=3, k=(3, 3, 3)) == NARROW                       # decoder conv_out
    assert r(Cin=16, Cout=512, k=(3, 3, 3)) == IGEMM                       # decoder conv_in
    assert r(Cin=8, Cout=128, k=(3, 3, 3)) == IGEMM                        # encoder conv_in (RGB padded to 8)
    assert r(Cin=16, Cout=128, k=(1, 1, 1)) == IGEMM                       # SpatialNorm conv_y / conv_b tables
    assert r(Cin=512, Cout=32, k=(3, 3, 3)) == IGEMM                       # encoder conv_out
    # the committed tiny fixture never leaves the register-staged kernel (why this file exists)
    assert r(Cin=32, Cout=32, k=(3, 3, 3)) == IGEMM and r(Cin=16, Cout=8, k=(3, 3, 3)) == IGEMM


def test_default_width_decode_small_vs_oracle(vae_default, gpu, fast_oracle_convs):
    """[1,16,5,8,12] -> 17 frames 64x96: chunk (0,3) with frame-0 replication, chunk (3,5) with the conv caches, then T == 1."""
    vae, cfg, sdf = vae_default
    g = torch.Generator(device=gpu).manual_seed(5)
    z = torch.randn(1, 16, 5, 8, 12, device=gpu, generator=g).to(BF)
    dec = vae.decode(z).sample
    assert dec.shape == (1, 3, 17, 64, 96) and dec.dtype == BF
    con, ex = _oracle_decode(sdf, cfg, z)
    _check_deep(dec, con, ex, "default-width decode, 17 frames 64x96 (2 chunks)")
    d1 = vae.decode(z[:, :, :1].contiguous()).sample
    c1, e1 = _oracle_decode(sdf, cfg, z[:, :, :1])
    _check_deep(d1, c1, e1, "default-width decode, T = 1")
    assert torch.equal(vae.decode(z).sample, dec)                          # caches cleared: re-entrant, bit-repeatable
    # batch of 2 (the layout kernels and the GroupNorm statistics are per batch item)
    z2 = torch.cat([z, z.flip(3)], 0)
    d2 = vae.decode(z2).sample
    assert torch.equal(d2[:1], dec)
    assert torch.equal(d2[1:], vae.decode(z.flip(3).contiguous()).sample)


@pytest.mark.parametrize("T,h,w", [(3, 17, 25), (4, 9, 33), (2, 30, 11)])
def test_default_width_decode_odd_sizes_vs_oracle(vae_default, gpu, fast_oracle_convs, T, h, w):
    """Odd latent grids (136x200, 72x264, 240x88 px: ragged position tiles at every stage, W not a multiple of anything) and even
    latent frame counts (T = 4 -> chunks (0,2), (2,4): the first chunk has an EVEN frame count, so the temporal upsample takes the
    all-frames branch, diffusers CogVideoXUpsample3D; T = 2 -> one chunk) through the default-width decoder against the oracle."""
    vae, cfg, sdf = vae_default
    g = torch.Generator(device=gpu).manual_seed(100 * T + h)
    z = torch.randn(1, 16, T, h, w, device=gpu, generator=g).to(BF)
    dec = vae.decode(z).sample
    con, ex = _oracle_decode(sdf, cfg, z)
    assert dec.shape == con.shape and dec.shape[3:] == (8 * h, 8 * w)
    _check_deep(dec, con, ex, f"default-width decode, latent [{T},{h},{w}]")
    assert torch.equal(vae.decode_to_frames(z), (dec / 2 + 0.5).clamp(0, 1).float())


@pytest.mark.parametrize("Fr,H,W", [(8, 40, 56), (6, 72, 40), (2, 136, 200)])
def test_default_width_encode_even_frame_counts_vs_oracle(vae_default, gpu, fast_oracle_convs, Fr, H, W):
    """Frame counts that are not 4 k + 1: 8 -> chunks of 4 + 4 (an EVEN first chunk: both temporal average pools take the all-pairs
    branch), 6 -> one chunk of 6 (remainder folded in: 6 -> 3 -> 1 + 1), 2 -> no full chunk at all (the reference's loop runs zero
    times and fails on an empty concat, :1199-1210: same error surface here); odd spatial grids."""
    vae, cfg, sdf = vae_default
    g = torch.Generator(device=gpu).manual_seed(Fr * H)
    x = (torch.rand(1, 3, Fr, H, W, device=gpu, generator=g) * 2 - 1).to(BF)
    if Fr // 4 == 0:
        with pytest.raises((RuntimeError, ValueError)):
            vae.encode(x)
        return
    post = vae.encode(x).latent_dist
    con, ex = _oracle_encode(sdf, cfg, x)
    assert post.mean.shape == con.mean.shape and post.mean.shape[3:] == (H // 8, W // 8)
    _check_deep(post.mean, con.mean, ex.mean, f"default-width encode mean, {Fr} frames {H}x{W}")
    _check_deep(post.logvar, con.logvar, ex.logvar, f"default-width encode logvar, {Fr} frames {H}x{W}")


def test_default_width_tiled_decode_vs_oracle(vae_default, gpu, fast_oracle_convs):
    """enable_tiling() at the default widths: 64 x 96 px tiles (8 x 12 latent) over a 16 x 20 latent -> 3 x 3 ragged tiles, two
    temporal chunks each, through the MFMA conv kernels; blended seams; against the oracle's restatement of tiled_decode."""
    vae, cfg, sdf = vae_default
    g = torch.Generator(device=gpu).manual_seed(15)
    z = torch.randn(1, 16, 5, 16, 20, device=gpu, generator=g).to(BF)
    kw = dict(tile_sample_min_height=64, tile_sample_min_width=96, tile_overlap_factor_height=0.25, tile_overlap_factor_width=0.25)
    try:
        vae.enable_tiling(**kw)
        dec = vae.decode(z).sample
    finally:
        vae.disable_tiling()
        vae.enable_tiling()                                  # back to the constructor's geometry ...
        vae.tile_sample_min_height, vae.tile_sample_min_width = cfg["sample_height"] // 2, cfg["sample_width"] // 2
        vae.tile_latent_min_height, vae.tile_latent_min_width = vae.tile_sample_min_height // 8, vae.tile_sample_min_width // 8
        vae.tile_overlap_factor_height, vae.tile_overlap_factor_width = 1 / 6, 1 / 5
        vae.disable_tiling()                                 # ... and off (module-scoped fixture)
    assert dec.shape == (1, 3, 17, 128, 160)
    with torch.no_grad():
        con = ovae.vae_tiled_decode(sdf, cfg, z.float(), prec="bf16", **kw).float()
        ex = ovae.vae_tiled_decode(sdf, cfg, z.float(), prec="fp32", **kw).float()
    _check_deep(dec, con, ex, "default-width tiled decode, 17 frames 128x160 (9 tiles)")


def test_default_width_encode_small_vs_oracle(vae_default, gpu):
    """[1,3,17,64,96] -> posterior over [1,16,5,8,12]: chunks of 5, 4, 4, 4 frames (remainder folded into the first, :1199-1205)."""
    vae, cfg, sdf = vae_default
    g = torch.Generator(device=gpu).manual_seed(6)
    x = (torch.rand(1, 3, 17, 64, 96, device=gpu, generator=g) * 2 - 1).to(BF)
    post = vae.encode(x).latent_dist
    assert post.mean.shape == (1, 16, 5, 8, 12)
    con, ex = _oracle_encode(sdf, cfg, x)
    _check_deep(post.mean, con.mean, ex.mean, "default-width encode mean, 17 frames 64x96 (4 chunks)")
    _check_deep(post.logvar, con.logvar, ex.logvar, "default-width encode logvar")
    p1 = vae.encode(x[:, :, :1].contiguous()).latent_dist
    c1, e1 = _oracle_encode(sdf, cfg, x[:, :, :1])
    _check_deep(p1.mean, c1.mean, e1.mean, "default-width encode, single frame")
    # odd sizes: 72x88 is divisible by 8 but not by the 512-position tile of the Cout = 128 stage; ragged tiles at every level
    xo = (torch.rand(1, 3, 5, 72, 88, device=gpu, generator=g) * 2 - 1).to(BF)
    po = vae.encode(xo).latent_dist
    co, eo = _oracle_encode(sdf, cfg, xo)
    _check_deep(po.mean, co.mean, eo.mean, "default-width encode, 5 frames 72x88 (ragged tiles)")


def test_fullsize_decode_first_two_chunks_vs_oracle(vae_default, gpu, fast_oracle_convs):
    """480x720: latent frames 0..4 of the 13-frame clip = the first two decode chunks (3 + 2 latent frames -> 9 + 8 frames), every
    element of [1,3,17,480,720] against the oracle on the device."""
    vae, cfg, sdf = vae_default
    g = torch.Generator(device=gpu).manual_seed(7)
    z = torch.randn(1, 16, 5, 60, 90, device=gpu, generator=g).to(BF)
    dec = vae.decode(z).sample
    assert dec.shape == (1, 3, 17, 480, 720)
    con, ex = _oracle_decode(sdf, cfg, z)
    torch.cuda.empty_cache()
    _check_deep(dec, con, ex, "480x720 decode, first two chunks (17 frames)")
    # the frames epilogue at size: fused (x/2+.5).clamp(0,1).float() == the torch expression on the bf16 output
    fr = vae.decode_to_frames(z)
    assert torch.equal(fr, (dec / 2 + 0.5).clamp(0, 1).float())


def test_fullsize_encode_first_two_chunks_vs_oracle(vae_default, gpu, fast_oracle_convs):
    """480x720: frames 0..8 = the first two encode chunks of the 49-frame clip (5 + 4 frames -> 2 + 1 latent frames), every
    element of the posterior moments [1,32,3,60,90]."""
    vae, cfg, sdf = vae_default
    g = torch.Generator(device=gpu).manual_seed(8)
    # smooth-ish content + noise, in [-1, 1] like a normalised video
    base = torch.rand(1, 3, 9, 30, 45, device=gpu, generator=g)
    x = torch.nn.functional.interpolate(base, size=(9, 480, 720), mode="trilinear", align_corners=False)
    x = ((x + 0.1 * torch.randn(x.shape, device=gpu, generator=g)).clamp(0, 1) * 2 - 1).to(BF)
    post = vae.encode(x).latent_dist
    assert post.mean.shape == (1, 16, 3, 60, 90)
    con, ex = _oracle_encode(sdf, cfg, x)
    torch.cuda.empty_cache()
    _check_deep(post.mean, con.mean, ex.mean, "480x720 encode mean, first two chunks (9 frames)")
    _check_deep(post.logvar, con.logvar, ex.logvar, "480x720 encode logvar")
    assert torch.equal(vae.encode(x).latent_dist.mean, post.mean)
