"""Orbit camera poses (BASELINE configs[3], reference inference_orbits.py:248-300 -> demo.py:538-586 -> models/utils.py:83-158).
CPU: the oracle's frame-by-frame restatement is pinned bit for bit by tests/golden/orbit_poses.safetensors (the reference's own
`generate_traj_specified`, run by tests/golden/make_golden.py poses); the product's all-frames-at-once `driver.orbit_poses`
is compared with the oracle's `get_poses_target` (fp32 trigonometry evaluated on vectors vs scalars: <= 1e-6 absolute on
entries of magnitude <= 5)."""
import torch

from oracle import poses as op

# the literal list make_golden.py used (variant 8 exercises d_r / d_x / d_y, which the orbit set leaves at 0)
VARIANTS = [[0, -30, 1.0, 0, 0], [0, 30, 1.0, 0, 0], [30, 0, 1.0, 0, 0], [0, -45, 1.0, 0, 0], [0, 45, 1.0, 0, 0],
            [45, 0, 1.0, 0, 0], [0, -90, 1.0, 0, 0], [0, 90, 1.0, 0, 0], [10, -20, 0.5, 0.3, -0.2]]
C2W_INIT = torch.tensor([[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]).unsqueeze(0)


def test_oracle_poses_match_reference_fixture_bitwise(golden):
    t, meta = golden("orbit_poses.safetensors")
    assert torch.equal(t["variants"], torch.tensor(VARIANTS, dtype=torch.float32))
    for radius in (1.0, 2.5, 5.0):
        want = t[f"poses_r{radius}"]
        assert want.shape == (9, 49, 4, 4)
        for i, (th, ph, dr, dx, dy) in enumerate(VARIANTS):
            got = op.generate_traj_specified(C2W_INIT, th, ph, dr * radius, dx, dy, 49)
            assert torch.equal(got, want[i]), (radius, i, float((got - want[i]).abs().max()))


def test_product_orbit_poses_match_oracle():
    from trajectorycrafter_amd.driver import ORBIT_VARIANTS, orbit_poses
    # the variant table is the reference's list (inference_orbits.py:258-283), in its order
    assert [n for n, _ in ORBIT_VARIANTS] == ["left_-30", "right_30", "top_30", "left_-45", "right_45", "top_45", "left_-90", "right_90"]
    assert [list(p) for _, p in ORBIT_VARIANTS] == [v[:2] + [1] + v[3:] for v in VARIANTS[:8]]
    for centre_depth, scale in ((1.0, 1.0), (2.5, 1.0), (7.0, 1.0), (2.0, 0.5)):          # 7.0 -> clamped to 5 (demo.py:543)
        depths = torch.full((3, 1, 6, 10), 9.0)
        depths[0, 0, 3, 5] = centre_depth
        for v in VARIANTS:
            for anchor in (0, 7):
                ps, pt, K = orbit_poses(depths, v, 49, radius_scale=scale, anchor_idx=anchor)
                os_, ot, oK = op.get_poses_target(depths, v, 49, radius_scale=scale, anchor_idx=anchor)
                assert ps.shape == pt.shape == (49, 4, 4) and K.shape == (49, 3, 3) and pt.dtype == torch.float32
                assert torch.equal(K, oK)
                assert float((pt - ot).abs().max()) <= 1e-6 and float((ps - os_).abs().max()) <= 1e-6
                assert torch.equal(ps, pt[anchor:anchor + 1].repeat(49, 1, 1))
    # frame 0 of every orbit is the anchor camera at distance `radius` in front of the scene centre
    ps, pt, _ = orbit_poses(torch.full((1, 1, 4, 4), 2.0), VARIANTS[6], 49)
    assert torch.allclose(pt[0], torch.tensor([[-1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 2.0], [0, 0, 0, 1]]))


def test_traj_poses_match_reference_fixture_and_oracle(golden, tmp_path):
    """`camera == 'traj'` (demo.py:566-573, models/utils.py:161-210): the oracle's `generate_traj_txt` is pinned bit for bit by the
    reference's own output on the key values of its two trajectory files (+ a short, linearly interpolated one); the product's
    all-frames-at-once `driver.traj_poses` agrees with the oracle's `get_poses_traj` to fp32 trigonometry (<= 2e-6)."""
    from trajectorycrafter_amd.driver import read_traj_txt, traj_poses
    t, _ = golden("orbit_poses.safetensors")
    for name in ("short", "loop1", "loop2"):
        th, ph, r = (t[f"traj_{name}_keys_{k}"].tolist() for k in ("theta", "phi", "r"))
        got = op.generate_traj_txt(C2W_INIT, ph, th, [v * 2.0 for v in r], 49)
        assert torch.equal(got, t[f"traj_{name}_poses"]), name
        for centre_depth, scale, anchor in ((2.0, 1.0, 0), (7.0, 1.0, 3), (3.0, 0.5, 0)):
            depths = torch.full((2, 1, 6, 10), 9.0)
            depths[0, 0, 3, 5] = centre_depth
            ps, pt, K = traj_poses(depths, th, ph, r, 49, radius_scale=scale, anchor_idx=anchor)
            os_, ot, oK = op.get_poses_traj(depths, th, ph, r, 49, radius_scale=scale, anchor_idx=anchor)
            assert pt.shape == (49, 4, 4) and pt.dtype == torch.float32 and torch.equal(K, oK)
            assert float((pt - ot).abs().max()) <= 2e-6 and float((ps - os_).abs().max()) <= 2e-6
    f = tmp_path / "traj.txt"
    f.write_text("0 2 10 15\n0 -3 -10\n0 0.02 0.09 0.16 0.25\n")
    assert read_traj_txt(str(f)) == ([0.0, 2.0, 10.0, 15.0], [0.0, -3.0, -10.0], [0.0, 0.02, 0.09, 0.16, 0.25])
