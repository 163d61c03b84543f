"""Point-cloud render: HIP `Warper.forward_warp` vs the oracle and the reference fixture.  GPU only.

fp32 float atomics commute only up to rounding, and a source pixel whose projected position is within an ulp of an
integer may or may not touch the neighbouring target pixel with a ~1e-7 weight.  A target pixel reached only by
corner weights <~1e-2 is ill-conditioned in fp32 for ANY implementation: positions ~1e2 px carry 1e-5 px of rounding,
i.e. a 1e-3 relative change of such a weight.  So: flow rtol 5e-5 / atol 1e-4 everywhere; values rtol 5e-5 / atol
1e-4 on >= 99.9 % of the pixels both sides hit and within 2 % of the value range on the rest; the hit masks may differ
on a stated fraction of the pixels (0 at fixture size)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import warp as owarp


@pytest.fixture(scope="module")
def warper():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from trajectorycrafter_amd.models.utils import Warper
    return Warper(device="cuda:0")


def _cmp(got, want, max_mask_diff):
    warped, mask2, wdepth, flow = (t.cpu() for t in got)
    ew, em, ed, ef = want
    torch.testing.assert_close(flow, ef, rtol=5e-5, atol=1e-4)
    diff = mask2 != em
    assert float(diff.float().mean()) <= max_mask_diff, float(diff.float().mean())
    same = ~diff
    for a, e in ((warped[same.expand_as(warped)], ew[same.expand_as(ew)]), (wdepth[same], ed[same])):
        err = (a - e).abs()
        bad = err > 1e-4 + 5e-5 * e.abs()
        assert float(bad.float().mean()) <= 1e-3, float(bad.float().mean())
        assert float((err / (1 + e.abs())).max()) <= 2e-2, float((err / (1 + e.abs())).max())
    assert float(warped.min()) >= -1.0 and float(warped.max()) <= 1.0


def test_forward_warp_matches_reference_fixture(warper, golden):
    t, _ = golden("warp_tiny.safetensors")
    got = warper.forward_warp(t["frame"], None, t["depth"], t["t1"], t["t2"], t["K"], None, False, twice=False)
    _cmp(got, (t["warped"], t["mask2"], t["warped_depth"], t["flow"]), 0.0)


def _scene(b, h, w, seed, with_mask):
    g = torch.Generator().manual_seed(seed)
    frame = torch.rand(b, 3, h, w, generator=g) * 2 - 1
    depth = 1.5 + 3 * torch.rand(b, 1, h, w, generator=g)
    depth[:, :, h // 4: h // 2, w // 3: w // 2] = 0.7
    k = torch.tensor([[0.7 * w, 0, w / 2], [0, 0.7 * w, h / 2], [0, 0, 1]])[None].repeat(b, 1, 1)
    t1 = torch.eye(4)[None].repeat(b, 1, 1)
    t2 = t1.clone()
    ang = torch.linspace(-0.2, 0.25, b)
    t2[:, 0, 0], t2[:, 0, 2], t2[:, 2, 0], t2[:, 2, 2] = ang.cos(), ang.sin(), -ang.sin(), ang.cos()
    t2[:, 0, 3] = torch.linspace(-0.4, 0.4, b)
    t2[:, 2, 3] = torch.linspace(0.3, -0.3, b)
    t1[:, 1, 3] = 0.1
    mask1 = (torch.rand(b, 1, h, w, generator=g) > 0.2).float() if with_mask else None
    return frame, mask1, depth, t1, t2, k


@pytest.mark.parametrize("b,h,w,with_mask", [(3, 37, 53, False), (2, 64, 96, True), (1, 1, 1, False)])
def test_forward_warp_matches_oracle(warper, b, h, w, with_mask):
    frame, mask1, depth, t1, t2, k = _scene(b, h, w, 11 * h + w, with_mask)
    k2 = k.clone()
    k2[:, 0, 0] *= 1.1
    want = owarp.forward_warp(frame, mask1, depth, t1, t2, k, k2)
    got = warper.forward_warp(frame, mask1, depth, t1, t2, k, k2, False, twice=False)
    _cmp(got, want, 1e-3 if h * w > 1 else 0.0)


def test_forward_warp_twice_matches_reference_fixture_and_oracle(warper, golden):
    """forward_warp(twice=True) (reference :294-347; its caller is notebooks/15_10_25_depth/collect_dataset.py): the fused first
    stage + three `tcx_bilinear_splat` launches against the reference's own output (fixture) and the oracle on a larger scene with a
    source mask.  Every stage re-splats the previous stage's output, so its ill-conditioned pixels compound: the shares of
    test_forward_warp_matches_oracle are allowed twice over."""
    t, _ = golden("warp_tiny.safetensors")
    f, m, d, none = warper.forward_warp(t["frame"], None, t["depth"], t["t1"], t["t2"], t["K"], None, False, twice=True)
    assert none is None and f.shape == t["twice_frame"].shape
    mdiff = (m.cpu() != t["twice_mask"])
    assert float(mdiff.float().mean()) <= 2e-3
    same = ~mdiff
    for a, e in ((f.cpu()[same.expand_as(f)], t["twice_frame"][same.expand_as(f)]), (d.cpu()[same], t["twice_depth"][same])):
        err = (a - e).abs()
        assert float((err > 1e-4 + 5e-5 * e.abs()).float().mean()) <= 2e-3 and float((err / (1 + e.abs())).max()) <= 4e-2
    frame, mask1, depth, t1, t2, k = _scene(2, 48, 80, 77, True)
    want = owarp.forward_warp_twice(frame, mask1, depth, t1, t2, k)
    got = warper.forward_warp(frame, mask1, depth, t1, t2, k, None, False, twice=True)
    mdiff = got[1].cpu() != want[1]
    assert float(mdiff.float().mean()) <= 4e-3
    same = ~mdiff
    for a, e in ((got[0].cpu()[same.expand_as(want[0])], want[0][same.expand_as(want[0])]), (got[2].cpu()[same], want[2][same])):
        err = (a - e).abs()
        assert float((err > 1e-4 + 5e-5 * e.abs()).float().mean()) <= 1e-2, float((err > 1e-4 + 5e-5 * e.abs()).float().mean())
        assert float(torch.quantile(err / (1 + e.abs()), 0.999)) <= 4e-2
    with pytest.raises(NotImplementedError, match="mask=False"):
        warper.forward_warp(frame, mask1, depth, t1, t2, k, None, True, twice=True)


@pytest.mark.parametrize("c,is_image,with_mask,scale", [(3, True, False, 1.0), (1, False, True, -1.0), (2, False, False, 1.0), (4, True, True, -1.0)])
def test_bilinear_splat_matches_oracle(c, is_image, with_mask, scale):
    """`tcx_bilinear_splat` = one `Warper.bilinear_splatting` (reference :422-583) with a given flow: 1..4 channels, image / non-image
    hole value and clamp, optional source mask, the negated flow of the return splat."""
    from trajectorycrafter_amd import ops
    g = torch.Generator().manual_seed(10 * c + int(is_image))
    b, h, w = 2, 40, 56
    src = torch.rand(b, c, h, w, generator=g) * 2 - 1
    depth = 0.5 + 3 * torch.rand(b, h, w, generator=g)
    flow = (torch.rand(b, 2, h, w, generator=g) - 0.5) * 9
    mask1 = (torch.rand(b, 1, h, w, generator=g) > 0.2).float() if with_mask else None
    want, wmask = owarp.bilinear_splat(src, mask1, depth, flow * scale, is_image)
    got, gmask = ops.bilinear_splat(src.cuda(), None if mask1 is None else mask1.cuda(), depth.cuda(), flow.cuda(), is_image, flow_scale=scale)
    mdiff = gmask.cpu() != wmask
    assert float(mdiff.float().mean()) <= 1e-3
    same = (~mdiff).expand_as(want)
    err = (got.cpu()[same] - want[same]).abs()
    assert float((err > 1e-4 + 5e-5 * want[same].abs()).float().mean()) <= 2e-3 and float(err.max()) <= 5e-2
    with pytest.raises(ops.TcxError):
        ops.bilinear_splat(torch.zeros(1, 5, 4, 4, device="cuda"), None, torch.ones(1, 4, 4, device="cuda"), torch.zeros(1, 2, 4, 4, device="cuda"), True)


def test_warper_public_helpers(warper):
    """`Warper.bilinear_splatting` (depth [b,1,h,w] or [b,h,w], flow12_mask multiplying the weights), `create_grid`,
    `camera_intrinsic_transform`, `get_device` — the reference's helper surface (models/utils.py:422-583, 628-682)."""
    from trajectorycrafter_amd.models.utils import Warper
    g = torch.Generator().manual_seed(2)
    b, h, w = 1, 24, 32
    src = torch.rand(b, 3, h, w, generator=g) * 2 - 1
    depth = 1 + torch.rand(b, 1, h, w, generator=g)
    flow = (torch.rand(b, 2, h, w, generator=g) - 0.5) * 5
    m1 = (torch.rand(b, 1, h, w, generator=g) > 0.3).float()
    fm = (torch.rand(b, 1, h, w, generator=g) > 0.3).float()
    got, gm = warper.bilinear_splatting(src, m1, depth, flow, fm, is_image=True)
    want, wm = owarp.bilinear_splat(src, m1 * fm, depth[:, 0], flow, True)
    assert float((gm.cpu() != wm).float().mean()) <= 2e-3
    same = (gm.cpu() == wm).expand_as(want)
    assert float((got.cpu()[same] - want[same]).abs().max()) <= 5e-2
    grid = Warper.create_grid(2, 3, 4)
    assert grid.shape == (2, 2, 3, 4) and grid[1, 0, 2].tolist() == [0, 1, 2, 3] and grid[0, 1, :, 0].tolist() == [0, 1, 2]
    k = Warper.camera_intrinsic_transform(1280, 720, (10, 20))
    assert k.shape == (4, 4) and k[0, 0] == 2100 and k[0, 2] == 620.0 and k[1, 2] == 350.0
    assert Warper.get_device("cpu").type == "cpu" and Warper.get_device("gpu0") == torch.device("cuda:0")


def test_forward_warp_mask_true_cleans_points(warper):
    """forward_warp(mask=True): holes dilated 5x5 inside the resolve kernel (reference clean_points :585-626)."""
    frame, _, depth, t1, t2, k = _scene(1, 60, 90, 21, False)
    want = owarp.forward_warp(frame, None, depth, t1, t2, k, mask=True)
    got = warper.forward_warp(frame, None, depth, t1, t2, k, None, True, twice=False)
    _cmp(got, tuple(t.float() for t in want), 1e-3)
    plain = warper.forward_warp(frame, None, depth, t1, t2, k, None, False, twice=False)
    assert float(got[1].mean()) < float(plain[1].mean())       # dilation removed pixels
    assert bool(((got[0] == -1).all(dim=1, keepdim=True) | (got[1] == 1)).all())


def test_forward_warp_per_frame_equals_batch1_calls(warper):
    """per_frame=True: one launch == b reference calls with batch 1 (demo.py:100-116 renders frame by frame)."""
    frame, mask1, depth, t1, t2, k = _scene(4, 48, 80, 99, True)
    depth[2] *= 3.0                                            # different max depth per item: the normaliser matters
    got = warper.forward_warp(frame, mask1, depth, t1, t2, k, None, False, twice=False, per_frame=True)
    parts = [owarp.forward_warp(frame[i:i + 1], mask1[i:i + 1], depth[i:i + 1], t1[i:i + 1], t2[i:i + 1], k[i:i + 1]) for i in range(4)]
    want = tuple(torch.cat([p[j] for p in parts]) for j in range(4))
    _cmp(got, want, 1e-3)
    batch = owarp.forward_warp(frame, mask1, depth, t1, t2, k)
    assert float((batch[0] - want[0]).abs().max()) > 1e-3      # and the batch-wide normaliser would have differed


def test_forward_warp_full_size_properties(warper):
    """49 frames of 576x1024 (the reference renders at this size, inference.py:41-42) in ONE call.  Properties that
    need no oracle: identity pose returns the input; three of the frames agree with batch-1 oracle calls."""
    b, h, w = 49, 576, 1024
    frame, _, depth, t1, t2, k = _scene(b, h, w, 3, False)
    dev = warper.device
    # identity on a fronto-parallel plane (with ragged depth a 1e-4-px rounding of the position lets a much nearer
    # neighbour win through its exp(50 ...) weight — in the reference as well)
    flat = torch.full_like(depth, 2.0).to(dev)
    out, mask, wd, flow = warper.forward_warp(frame.to(dev), None, flat, t1, t1, k, None, False, twice=False)
    assert float(flow.abs().max()) < 2e-3 and float(mask.mean()) == 1.0
    torch.testing.assert_close(out.cpu(), frame, rtol=0, atol=2e-3)
    got = warper.forward_warp(frame.to(dev), None, depth.to(dev), t1, t2, k, None, False, twice=False, per_frame=True)
    for i in (0, 23, 48):
        sl = slice(i, i + 1)
        want = owarp.forward_warp(frame[sl], None, depth[sl], t1[sl], t2[sl], k[sl])
        _cmp(tuple(g[sl] for g in got), want, 1e-4)
    assert 0.5 < float(got[1].mean()) < 1.0
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    fd, dd = frame.to(dev), depth.to(dev)
    ev[0].record()
    for _ in range(5):
        warper.forward_warp(fd, None, dd, t1, t2, k, None, False, twice=False, per_frame=True)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"\nforward_warp 49x576x1024: {ev[0].elapsed_time(ev[1]) / 5:.3f} ms/call")
