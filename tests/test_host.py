"""CPU tests of the host-side pieces (liblins_host.so): synthetic scan generator, feature
front-end, transformToEnd, StatePredictor mirror."""
import ctypes as C

import numpy as np


def test_synthetic_pairs_are_deterministic_and_distinct(host):
    a, b, c = host.synth_pair(5), host.synth_pair(5), host.synth_pair(6)
    for name in ("surf_flat", "corner_sharp", "surf_last", "corner_last"):
        assert np.array_equal(getattr(a, name), getattr(b, name))
    assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov)
    assert a.sizes() != c.sizes() or not np.array_equal(a.state, c.state)
    assert a.bytes_per_iter() == 16 * sum(a.sizes()) + 8 * 19 + 8 * 28


def test_feature_clouds_respect_the_reference_caps_and_layout(pairs):
    for p in pairs:
        n_sharp, n_flat, n_ls, n_lf = p.sizes()
        assert 0 < n_sharp <= 192 and 0 < n_flat <= 144 and n_sharp <= n_ls <= 1920  # SE:727-793
        assert 1000 < n_lf < 28800
        for cloud in (p.surf_flat, p.corner_sharp, p.surf_last, p.corner_last):
            assert np.isfinite(cloud).all()
            ring = cloud[:, 3].astype(int)
            assert ring.min() >= 0 and ring.max() <= 15 and (np.diff(ring) >= 0).all()  # ring-sorted
            frac = cloud[:, 3] - ring
            assert frac.min() > -0.03 and frac.max() < 0.13  # SCAN_PERIOD * relTime, quirks included
        assert p.surf_flat[:, 3].astype(int).max() <= 6  # flat points come from the ground rings (IP:248-267)
        rng = np.linalg.norm(p.surf_last[:, :3], axis=1)
        assert rng.min() > 0.5 and rng.max() < 60
        # prior: unit quaternion, symmetric PSD covariance, gravity norm 9.81
        assert abs(np.linalg.norm(p.state[6:10]) - 1) < 1e-12
        assert np.allclose(p.cov, p.cov.T, atol=1e-18) and np.linalg.eigvalsh(p.cov).min() > -1e-15
        assert abs(np.linalg.norm(p.state[16:19]) - 9.81) < 1e-9
        # the IMU prior is close to the simulated motion
        assert np.linalg.norm(p.state[:3] - p.meta["true_t"]) < 0.05


def test_frontend_on_raw_scan_matches_the_pair(host):
    raw = host.synth_raw_scan(2, 1)
    assert 20000 < len(raw) <= 28800
    f = host.frontend_extract(raw)
    pair = host.synth_pair(2)
    assert np.array_equal(f["corner_sharp"], pair.corner_sharp)
    assert np.array_equal(f["surf_flat"], pair.surf_flat)
    assert f["n_segmented"] > 5000
    # the sharp points are a subset of the less-sharp ones (SE:749-757)
    ls = {tuple(r) for r in f["corner_less_sharp"]}
    assert all(tuple(r) in ls for r in f["corner_sharp"])


def test_transform_to_end_inverts_the_full_motion(host, oracle, pkg):
    rng = np.random.default_rng(0)
    t = np.array([0.6, -0.05, 0.02])
    q = oracle.axis2quat([0.01, -0.02, 0.05])
    pts = np.concatenate([rng.uniform(-20, 20, (200, 3)), (rng.integers(0, 16, 200) + rng.uniform(0, 0.1, 200))[:, None]], 1)
    pts = pts.astype(np.float32)
    end = host.transform_to_end(t, q, pts)
    # transformToEnd = (full pose)^-1 o transformToStart  (SE:1083-1101)
    st = np.zeros(19)
    st[0:3], st[6:10] = t, q
    start = oracle.transform_to_start(pkg.default_params(), st, pts)
    qc = np.array([q[0], -q[1], -q[2], -q[3]])

    def rot(qq, v):
        w, x, y, z = qq
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        return v @ R.T

    want = rot(qc, start[:, :3].astype(np.float64) - t)
    assert np.abs(end[:, :3] - want).max() < 1e-4  # start is already f32-rounded
    assert np.array_equal(end[:, 3], pts[:, 3])


def test_state_predictor_mirror(host):
    L = host.lib()
    prm = host.FilterParams()
    L.lins_filter_default_params(C.byref(prm))
    f = host.Filter()
    v = (C.c_double * 3)(2.0, 0.0, 0.0)
    z = (C.c_double * 3)(0.0, 0.0, 0.0)
    L.lins_filter_init(C.byref(f), C.byref(prm), v, z, z)
    cov0 = np.array(f.cov[:]).reshape(18, 18)
    assert cov0[0, 0] == 0 and cov0[9, 9] == 1e-4 and cov0[11, 11] == 4e-4 and cov0[15, 15] == 0.01  # yaml:34-62
    acc = (C.c_double * 3)(0.0, 0.0, 9.81)  # at rest the accelerometer reads +g
    gyr = (C.c_double * 3)(0.0, 0.0, 0.1)
    for _ in range(40):
        L.lins_filter_predict(C.byref(f), 0.0025, acc, gyr)
    s = np.array(f.state[:])
    assert abs(f.time - 0.1) < 1e-12
    assert np.allclose(s[0:3], [0.2, 0.0, 0.0], atol=2e-3)  # constant velocity, gravity cancels
    assert np.allclose(s[6:10], [np.cos(0.005), 0, 0, np.sin(0.005)], atol=1e-9)  # 0.1 rad/s * 0.1 s
    cov = np.array(f.cov[:]).reshape(18, 18)
    assert np.allclose(cov, cov.T) and np.linalg.eigvalsh(cov).min() > -1e-15 and cov[0, 0] > 0 and cov[3, 3] > 0
    L.lins_filter_reset1(C.byref(f))  # KF:320-352
    s1 = np.array(f.state[:])
    c1 = np.array(f.cov[:]).reshape(18, 18)
    assert np.array_equal(s1[0:3], [0, 0, 0]) and np.array_equal(s1[6:10], [1, 0, 0, 0])
    assert abs(np.linalg.norm(s1[3:6]) - np.linalg.norm(s[3:6])) < 1e-12  # velocity rotated into the new frame
    assert c1[0, 0] == 0 and c1[6, 6] == 0 and np.allclose(c1[0:3, 3:6], 0)  # cross terms dropped
    assert np.allclose(np.trace(c1[3:6, 3:6]), np.trace(cov[3:6, 3:6]))


def test_open_scene_family_is_seeded_sparse_and_leaves_the_room_alone(host):
    """The second scene family of the generator (csrc/host/synth.cpp, scene 1: open ground, trunks, far wall segments,
    30 % of the returns lost, a moving box): deterministic in (seed, index), far sparser than the room and still
    corner-rich, its truth consistent with its own motion — and asking for it changes nothing about scene 0."""
    a, b = host.synth_pair(5, scene=1), host.synth_pair(5, scene=1)
    for f in ("surf_flat", "corner_sharp", "surf_last", "corner_last", "state", "cov"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    room = host.synth_pair(5)
    assert np.array_equal(room.surf_last, host.synth_pair(5, scene=0).surf_last)
    assert len(a.surf_last) < 0.4 * len(room.surf_last) and len(a.corner_sharp) > 60 and len(a.surf_flat) > 60
    assert a.meta["n_raw"][0] < 0.6 * room.meta["n_raw"][0]  # (lost returns and sky)
    # ring-sorted targets with ring ids < 16 (what the grid kernels need), queries never outnumber the targets
    for cloud in (a.surf_last, a.corner_last):
        rings = cloud[:, 3].astype(int)
        assert (np.diff(rings) >= 0).all() and rings.max() < 16
    assert len(a.surf_flat) <= len(a.surf_last) and len(a.corner_sharp) <= len(a.corner_last)
    raw = host.synth_raw_scan(5, 1, scene=1)
    assert len(raw) == a.meta["n_raw"][1] and np.isfinite(raw).all()
