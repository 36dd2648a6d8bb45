"""Edge cases of the HIP path vs the oracle: adversarial clouds (exact distance ties,
duplicates, lattice points, points near the sensor axis), empty / ragged / maximum-size
inputs, inputs that violate the binned-search precondition (unsorted rings, ring ids
>= 16 -> exact brute fallback), the forward-walk query-count quirk (SE:859, 983),
ICP_FREQ > 1, divergence + ICP fallback, and the C ABI's error behaviour."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_pair(pkg, rng, n_sf, n_cs, n_sl, n_cl, kind="lattice", rings=16, sort=True, state=None):
    """Random scan pair whose targets sit on a 0.25 m lattice (=> many exact f32 ties)."""

    def cloud(n, lattice):
        if n == 0:
            return np.zeros((0, 4), np.float32)
        az = rng.uniform(-np.pi, np.pi, n)
        rg = rng.uniform(0.3 if kind == "axis" else 2.0, 3.0 if kind == "axis" else 25.0, n)
        z = rng.uniform(-2.0, 3.0, n)
        xyz = np.stack([rg * np.cos(az), rg * np.sin(az), z], 1)
        if lattice:
            xyz = np.round(xyz * 4) / 4
        ring = rng.integers(0, rings, n)
        if kind == "dup":  # exact duplicates, possibly on different rings
            xyz[n // 2:] = xyz[: n - n // 2]
        if sort:
            ring = np.sort(ring)
        frac = rng.uniform(0.0, 0.1, n)
        return np.concatenate([xyz, (ring + frac)[:, None]], 1).astype(np.float32)

    if state is None:
        state = np.zeros(19)
        state[0:3] = rng.normal(0, 0.2, 3)
        q = np.array([1.0, *rng.normal(0, 0.01, 3)])
        state[6:10] = q / np.linalg.norm(q)
        state[3:6] = rng.normal(0, 1, 3)
        state[16:19] = [0, 0, -9.81]
    A = rng.normal(0, 1, (18, 18))
    cov = A @ A.T * 1e-4 + np.diag(rng.uniform(1e-6, 1e-3, 18))
    sl = cloud(n_sl, True)
    cl = cloud(n_cl, True)
    sf = cloud(n_sf, kind != "offgrid")
    cs = cloud(n_cs, kind != "offgrid")
    return pkg.ScanPair(sf, cs, sl, cl, state, cov)


def assert_same_corr(got, want, what):
    for f in ("ind1", "ind2", "ind3", "accepted"):
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, f"{what}.{f} differs at {bad[:8]}: got {got[f][bad[:8]]} want {want[f][bad[:8]]}"
    for f in ("coeff", "sel"):
        a = got[f].view(np.int32).astype(np.int64)
        b = want[f].view(np.int32).astype(np.int64)
        assert np.abs(a - b).max(initial=0) <= 1, f"{what}.{f}"


@pytest.fixture(scope="module")
def ctx(pkg, ieskf):
    c = ieskf.IeskfContext(pkg.default_params(num_iter=30), device=0, max_batch=8, max_targets=30000)
    yield c
    c.close()


@pytest.mark.parametrize("search", ["brute", "binned", "lds", "lds1", "mr", "auto"])
@pytest.mark.parametrize("kind", ["lattice", "dup", "axis", "offgrid"])
def test_adversarial_clouds_exact_indices(pkg, oracle, ctx, search, kind):
    ctx.set_search(search)
    prm = pkg.default_params()
    rng = np.random.default_rng({"lattice": 101, "dup": 202, "axis": 303, "offgrid": 404}[kind])
    for trial in range(4):
        n_sl = int(rng.integers(50, 3000))
        pair = make_pair(pkg, rng, int(rng.integers(1, 200)), int(rng.integers(1, 200)), n_sl,
                         int(rng.integers(5, 600)), kind)
        for it in (0, 1):
            surf, corner = ctx.correspondences(pair, pair.state, it)
            ws, wc = oracle.correspondences(prm, pair, pair.state, it, oracle.NN_BRUTE)
            assert_same_corr(surf, ws, f"{kind}/{trial}/it{it}/surf")
            assert_same_corr(corner, wc, f"{kind}/{trial}/it{it}/corner")


@pytest.mark.parametrize("search", ["brute", "binned", "lds", "lds1", "mr", "auto"])
def test_forward_walk_is_bounded_by_query_count(pkg, oracle, ctx, search):
    """SE:859/983: the forward walk stops at j < N_query.  Few queries => forward part empty;
    many queries (> targets) => our min(N_query, N_target) guard."""
    ctx.set_search(search)
    prm = pkg.default_params()
    rng = np.random.default_rng(7)
    for n_q, n_t in ((3, 800), (900, 300), (64, 64)):
        pair = make_pair(pkg, rng, n_q, n_q, n_t, n_t, "lattice")
        surf, corner = ctx.correspondences(pair, pair.state, 1)
        ws, wc = oracle.correspondences(prm, pair, pair.state, 1, oracle.NN_BRUTE)
        assert_same_corr(surf, ws, "surf")
        assert_same_corr(corner, wc, "corner")
        if n_q == 3:
            fwd = (ws["ind2"] > ws["ind1"]) & (ws["ind2"] >= 0)
            assert not fwd[ws["ind1"] >= 3].any()
        if n_q == 900:  # 1800 queries: several 336-slot rounds in the LDS kernel, reduced across rounds
            one = pkg.default_params(num_iter=1)
            _, tr = oracle.ieskf(one, pair, oracle.FORM_DENSE, oracle.NN_BRUTE, trace=True)
            sums, ms, mc = ctx.reduce_pass(pair, pair.state, 0)
            assert ms == int(tr["surf"][0]["accepted"].sum()) and mc == int(tr["corner"][0]["accepted"].sum())
            assert np.abs(sums - tr["sums28"][0]).max() <= 1e-10 * max(1.0, np.abs(tr["sums28"][0]).max())
            got = ctx.update(pair)
            want = oracle.ieskf(pkg.default_params(num_iter=30), pair, oracle.FORM_DENSE, oracle.NN_BRUTE)
            assert (got.iters, got.converged, got.diverged) == (want.iters, want.converged, want.diverged)
            if not want.diverged:
                assert np.abs(got.state - want.state).max() <= 1e-6 * max(1.0, np.abs(want.state).max())


@pytest.mark.parametrize("search", ["brute", "binned", "lds", "lds1", "mr", "auto"])
def test_mixed_batch_with_one_oversized_scan(pkg, oracle, ctx, search):
    """One scan too large for LDS sends the whole batch down the global-memory grid path."""
    ctx.set_search(search)
    rng = np.random.default_rng(21)
    pairs = [make_pair(pkg, rng, 40, 50, 600, 200, "offgrid"), make_pair(pkg, rng, 40, 50, 12000, 900, "offgrid"),
             make_pair(pkg, rng, 30, 30, 500, 100, "offgrid")]
    for got, pair in zip(ctx.update_batch(pairs), pairs):
        want = oracle.ieskf(pkg.default_params(num_iter=30), pair, oracle.FORM_DENSE, oracle.NN_BRUTE)
        assert (got.iters, got.converged, got.diverged, got.m_surf, got.m_corner) == (
            want.iters, want.converged, want.diverged, want.m_surf, want.m_corner)
        if not want.diverged:
            assert np.abs(got.state - want.state).max() <= 1e-6 * max(1.0, np.abs(want.state).max())


@pytest.mark.parametrize("search", ["brute", "binned", "lds", "lds1", "mr", "auto"])
def test_unsorted_rings_and_high_ring_ids_fall_back_exactly(pkg, oracle, ctx, search):
    ctx.set_search(search)
    prm = pkg.default_params()
    rng = np.random.default_rng(11)
    for rings, sort in ((16, False), (40, True), (64, False)):
        pair = make_pair(pkg, rng, 60, 60, 700, 300, "lattice", rings=rings, sort=sort)
        surf, corner = ctx.correspondences(pair, pair.state, 1)
        ws, wc = oracle.correspondences(prm, pair, pair.state, 1, oracle.NN_BRUTE)
        assert_same_corr(surf, ws, "surf")
        assert_same_corr(corner, wc, "corner")


@pytest.mark.parametrize("search", ["brute", "binned", "lds", "lds1", "mr", "auto"])
def test_empty_and_ragged_inputs(pkg, oracle, ctx, search):
    ctx.set_search(search)
    prm = pkg.default_params(num_iter=5)
    rng = np.random.default_rng(3)
    shapes = [(0, 0, 0, 0), (0, 10, 100, 50), (10, 0, 100, 50), (10, 10, 0, 50), (10, 10, 100, 0), (1, 1, 1, 1),
              (5, 7, 3, 2)]
    pairs = [make_pair(pkg, rng, *s) for s in shapes]
    with_ctx = ctx.update_batch(pairs)
    for got, pair in zip(with_ctx, pairs):
        want = oracle.ieskf(pkg.default_params(num_iter=30), pair, oracle.FORM_DENSE, oracle.NN_BRUTE)
        assert (got.iters, got.converged, got.diverged) == (want.iters, want.converged, want.diverged)
        assert (got.m_surf, got.m_corner) == (want.m_surf, want.m_corner)
        if not want.diverged:
            assert np.abs(got.state - want.state).max() <= 1e-6 * max(1.0, np.abs(want.state).max())
            assert np.abs(got.cov - want.cov).max() <= 1e-8 * np.abs(want.cov).max()
    del prm


@pytest.mark.parametrize("search", ["brute", "binned", "lds", "lds1", "mr", "auto"])
def test_maximum_sizes(pkg, oracle, ctx, search):
    """1024 queries per cloud (LINS_MAX_QUERY; > one 512-slot reduction round) and a
    full 16x1800 target cloud."""
    ctx.set_search(search)
    prm = pkg.default_params()
    rng = np.random.default_rng(5)
    pair = make_pair(pkg, rng, 1024, 1024, 28800, 1920, "offgrid")
    surf, corner = ctx.correspondences(pair, pair.state, 1)
    ws, wc = oracle.correspondences(prm, pair, pair.state, 1, oracle.NN_BRUTE)
    assert_same_corr(surf, ws, "surf")
    assert_same_corr(corner, wc, "corner")
    sums, ms, mc = ctx.reduce_pass(pair, pair.state, 1)
    assert ms == int(ws["accepted"].sum()) and mc == int(wc["accepted"].sum())


def test_icp_freq_reuses_indices(pkg, ieskf, oracle, pairs):
    prm = pkg.default_params(num_iter=12, icp_freq=3)
    with ieskf.IeskfContext(prm, max_batch=2, max_targets=16384, search="lds") as c:
        for pair in pairs[:2]:
            got = c.update(pair)
            want = oracle.ieskf(prm, pair, oracle.FORM_DENSE, oracle.NN_BRUTE)
            assert (got.iters, got.converged, got.diverged) == (want.iters, want.converged, want.diverged)
            assert np.abs(got.state[:3] - want.state[:3]).max() <= 1e-6
            assert np.abs(got.cov - want.cov).max() <= 1e-9 * np.abs(want.cov).max()


@pytest.mark.parametrize("search", ["brute", "binned", "lds", "lds1", "mr", "auto"])
def test_residual_blow_up_diverges_like_the_reference(pkg, ieskf, oracle, search):
    """SE:566-570 (diverged == 1): tests/diverging.py builds a pair whose second iteration's residual norm exceeds
    ten times the first's — one 1 mm row, then twenty 10 cm rows after the update jumped 10 m along the only free
    direction.  The C ABI reports diverged = 1, the iteration count of the reference's loop and the UN-updated
    filter state and covariance (SE:585-592 keeps Pk_)."""
    from diverging import make_diverging_pair

    pair = make_diverging_pair(pkg)
    prm = pkg.default_params(num_iter=30)
    want = oracle.ieskf(prm, pair, oracle.FORM_DENSE, oracle.NN_BRUTE)
    assert (want.iters, want.converged, want.diverged, want.m_surf, want.m_corner) == (2, 0, 1, 20, 0)
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384, search=search) as c:
        got = c.update(pair)
        assert (got.iters, got.converged, got.diverged) == (want.iters, want.converged, want.diverged)
        assert (got.m_surf, got.m_corner) == (want.m_surf, want.m_corner)
        assert abs(got.residual_norm - want.residual_norm) <= 1e-9
        assert np.array_equal(got.state, pair.state) and np.array_equal(got.cov, pair.cov)
        # performIESKF as the node sees it: the ICP fallback runs (SE:585-592) and the covariance stays the prior's
        full, used = c.perform_ieskf(pair)
        assert used
        wfull = oracle.perform_ieskf(prm, pair, oracle.FORM_DENSE, oracle.NN_BRUTE)
        assert np.abs(full.state[:3] - wfull.state[:3]).max() <= 1e-6
        assert np.abs(full.state[6:10] - wfull.state[6:10]).max() <= 1e-7
        assert np.array_equal(full.cov, pair.cov)


def test_icp_matches_oracle(pkg, ieskf, oracle, pairs):
    """estimateTransform (SE:1163-1320) driven through GPU correspondences == oracle ICP."""
    import ctypes as C

    prm = pkg.default_params(num_iter=30)
    defs = ieskf  # noqa: F841
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384, search="lds") as c:
        pair = pairs[1]
        # force the fallback: NaN covariance => NaN update => diverged = 2 (SE:552-563)
        bad = pkg.ScanPair(pair.surf_flat, pair.corner_sharp, pair.surf_last, pair.corner_last, pair.state,
                           np.full((18, 18), np.nan))
        full, used = c.perform_ieskf(bad)
        assert used
        t, q, _ = oracle.icp(prm, bad, pair.state[0:3], pair.state[6:10], oracle.NN_BRUTE)
        assert np.abs(full.state[0:3] - t).max() <= 1e-6
        assert np.abs(full.state[6:10] - q).max() <= 1e-7
    del C


def test_device_icp_matches_oracle_and_the_host_path(pkg, ieskf, oracle, host, monkeypatch):
    """the whole ICP fallback (SE:1163-1320) in one kernel per scan == the oracle's
    estimateTransform, from perturbed poses (so that several Gauss-Newton rounds run, the first with
    the degeneracy projection), and == the split path (device correspondences + host GN step)."""
    prm = pkg.default_params(num_iter=30)
    rng = np.random.default_rng(11)
    batch = []
    for p in host.synth_batch(12, start=300):
        st = p.state.copy()
        st[0:3] += rng.normal(0, 0.15, 3)
        dq = np.array([1.0, *rng.normal(0, 0.01, 3)])
        w, x, y, z = st[6:10]
        a, b, c, d = dq / np.linalg.norm(dq)
        st[6:10] = [w * a - x * b - y * c - z * d, w * b + x * a + y * d - z * c, w * c - x * d + y * a + z * b,
                    w * d + x * c - y * b + z * a]
        batch.append(pkg.ScanPair(p.surf_flat, p.corner_sharp, p.surf_last, p.corner_last, st, p.cov))
    with ieskf.IeskfContext(prm, max_batch=len(batch), max_targets=16384) as c:
        got = c.icp_update_batch(batch)
        rounds = []
        for pair, g in zip(batch, got):
            t, q, iters = oracle.icp(prm, pair, pair.state[0:3], pair.state[6:10], oracle.NN_KDTREE)
            assert g.iters == iters and g.diverged == 0
            assert np.abs(g.state[0:3] - t).max() <= 1e-6 and np.abs(g.state[6:10] - q).max() <= 1e-7
            keep = [3, 4, 5] + list(range(10, 19))
            assert np.array_equal(g.state[keep], pair.state[keep]) and np.array_equal(g.cov, pair.cov)
            rounds.append(iters)
        assert max(rounds) >= 2
        # the split path (what ineligible clouds take) lands on the same pose
        nan_cov = np.full((18, 18), np.nan)  # NaN covariance => the filter diverges => fallback (SE:552-563)
        bad = pkg.ScanPair(batch[0].surf_flat, batch[0].corner_sharp, batch[0].surf_last, batch[0].corner_last,
                           batch[0].state, nan_cov)
        dev, used = c.perform_ieskf(bad)
        assert used and dev.diverged == 2
        monkeypatch.setenv("LINS_ICP_HOST", "1")
        hst, used = c.perform_ieskf(bad)
        monkeypatch.delenv("LINS_ICP_HOST")
        assert used and np.abs(dev.state - hst.state).max() <= 1e-9
        assert np.abs(dev.state[0:3] - got[0].state[0:3]).max() <= 1e-12
    # ICP_FREQ > 1: triplets are reused between searches; corners are searched (and their triplets
    # replaced) only on rounds that accepted >= 10 plane rows (SE:1175-1178) — incl. a scan with too few
    few = pkg.ScanPair(batch[1].surf_flat[:6], batch[1].corner_sharp, batch[1].surf_last, batch[1].corner_last,
                       batch[1].state, batch[1].cov)
    for freq in (2, 3):
        p2 = pkg.default_params(num_iter=12, icp_freq=freq)
        with ieskf.IeskfContext(p2, max_batch=4, max_targets=16384) as c:
            for pair, g in zip(batch[:3] + [few], c.icp_update_batch(batch[:3] + [few])):
                t, q, iters = oracle.icp(p2, pair, pair.state[0:3], pair.state[6:10], oracle.NN_KDTREE)
                assert g.iters == iters
                assert np.abs(g.state[0:3] - t).max() <= 1e-6 and np.abs(g.state[6:10] - q).max() <= 1e-7
    # clouds off the grid are refused, not silently approximated
    off = pkg.ScanPair(batch[0].surf_flat, batch[0].corner_sharp, batch[0].surf_last[::-1].copy(), batch[0].corner_last,
                       batch[0].state, batch[0].cov)
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384) as c:
        with pytest.raises(ieskf.LinsError, match="-7"):
            c.icp_update_batch([off])


def test_abi_error_behaviour(pkg, ieskf):
    prm = pkg.default_params()
    rng = np.random.default_rng(1)
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=256) as c:
        good = make_pair(pkg, rng, 10, 10, 100, 100)
        with pytest.raises(ieskf.LinsError, match="-3"):  # capacity: batch of 2 into max_batch 1
            c.update_batch([good, good])
        with pytest.raises(ieskf.LinsError, match="-3"):  # capacity: too many targets
            c.update(make_pair(pkg, rng, 10, 10, 1000, 100))
        bad = make_pair(pkg, rng, 10, 10, 100, 100)
        bad.surf_last[5, 0] = np.nan
        with pytest.raises(ieskf.LinsError, match="-4"):
            c.update(bad)
        bad = make_pair(pkg, rng, 10, 10, 100, 100)
        bad.corner_last[3, 3] = 70.0  # ring id out of range
        with pytest.raises(ieskf.LinsError, match="-4"):
            c.update(bad)
        with pytest.raises(ieskf.LinsError, match="-6"):  # run before upload
            c.run()
        assert c.update(good).iters >= 1
    with pytest.raises(ieskf.LinsError, match="-1"):
        ieskf.IeskfContext(pkg.default_params(num_iter=0))
    with pytest.raises(ieskf.LinsError, match="-5"):
        ieskf.IeskfContext(prm, device=99)


def test_update_point_cloud_reprojection_matches_host(pkg, ieskf, host, pairs):
    """transformToEnd of whole clouds on device == the host restatement
    (SE:1083-1101), XYZ and the YZX copy (SE:1125-1129); empty and in-place clouds included."""
    rng = np.random.default_rng(9)
    clouds, poses = [], []
    for p in pairs[:3]:
        for cl in (p.surf_last, p.corner_last):
            clouds.append(cl)
            poses.append((rng.normal(0, 0.3, 3), (lambda v: v / np.linalg.norm(v))(np.array([1.0, *rng.normal(0, 0.02, 3)]))))
    clouds.append(np.zeros((0, 4), np.float32))
    poses.append((np.zeros(3), np.array([1.0, 0, 0, 0])))
    with ieskf.IeskfContext(pkg.default_params(), max_batch=4, max_targets=16384) as c:
        xyz, yzx = c.transform_to_end(clouds, poses, yzx=True)
        ms, nbytes = c.reproject_stats()
        assert ms > 0 and nbytes == 48 * sum(len(cl) for cl in clouds)
        for cl, (t, q), a, b in zip(clouds, poses, xyz, yzx):
            want = host.transform_to_end(t, q, cl)
            ulp = np.abs(a.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
            assert ulp.max(initial=0) <= 1 and (ulp > 0).mean() <= 1e-3 if ulp.size else True
            assert np.array_equal(b[:, 0], a[:, 1]) and np.array_equal(b[:, 1], a[:, 2]) and np.array_equal(b[:, 2], a[:, 0])
            assert np.array_equal(b[:, 3], cl[:, 3]) and np.array_equal(a[:, 3], cl[:, 3])
        xyz2, none = c.transform_to_end(clouds[:2], poses[:2], yzx=False)
        assert none == [None, None] and np.array_equal(xyz2[0], xyz[0])
        # a context that has just re-projected can go straight back to IESKF updates
        assert c.update(pairs[0]).iters >= 1


@pytest.mark.parametrize("search", ["lds", "lds1", "mr"])
@pytest.mark.parametrize("shape", [(0, 400, 300), (0, 900, 3000), (700, 0, 5000), (500, 500, 9500)])
def test_more_queries_than_one_round_of_lanes(pkg, oracle, ctx, search, shape):
    """More queries than a workgroup has query slots (336 / 384 / 512): several rounds over the same
    LDS grid, no state carried from round to round (regression: a 16/32-bit aliasing assumption in
    the grid build once let later rounds read stale cell bounds).  The last shape also exceeds the
    resident part of the multi-resident kernel's grid (hybrid LDS / global storage)."""
    ctx.set_search(search)
    prm = pkg.default_params()
    rng = np.random.default_rng(7)
    n_sq, n_cq, n_t = shape
    pair = make_pair(pkg, rng, n_sq, n_cq, n_t, n_t // 4 if n_t > 4000 else n_t, "lattice")
    for it in (0, 1):
        surf, corner = ctx.correspondences(pair, pair.state, it)
        ws, wc = oracle.correspondences(prm, pair, pair.state, it, oracle.NN_BRUTE)
        assert_same_corr(surf, ws, "surf")
        assert_same_corr(corner, wc, "corner")


def test_device_front_end_matches_the_host_restatement(pkg, ieskf, host):
    """undistortPcl .. extractFeatures (SE:619-827) on the device == the host
    restatement on the same segmented scans: identical picks (same points, same order, same counts),
    every field bit-equal (both sides use the fixed-sequence lins_atan2f for the time tags).  Includes the
    scans the IESKF fixtures are made of."""
    segs = [host.frontend_segment(host.synth_raw_scan(idx, k)) for idx in range(3) for k in (0, 1)]
    want = [host.frontend_extract_segmented(s) for s in segs]
    with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
        got = c.extract_features_batch(segs)
        ms, nbytes = c.frontend_stats()
        assert ms > 0 and nbytes > 0
        # and again on the same context: the buffers are reused
        again = c.extract_features_batch(segs[:2])
    for g, w in zip(got + again, want + want[:2]):
        for k in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"):
            assert g[k].shape == w[k].shape, (k, g[k].shape, w[k].shape)
            assert np.array_equal(g[k], w[k]), k  # coordinates AND time tags, bit for bit (shared lins_atan2f)
        assert g["n_segmented"] == w["n_segmented"] and g["n_outlier"] == w["n_outlier"]


@pytest.mark.parametrize("scene", [0, 1], ids=["room", "open"])
def test_device_resident_streams_reproduce_the_staged_path(pkg, ieskf, host, scene):
    """Two scans per stream through lins_streams_step (front-end -> update -> re-projection, clouds
    resident in HBM) == the staged path on the same data: host front-end, host re-projection of the
    first scan's clouds with the bootstrap pose, lins_ieskf_update_batch on the resulting pairs.  Both scene
    families of the generator (the open one: sparse rings, lost returns, a moving box)."""
    n = 6
    pairs = host.synth_batch(n, start=40, scene=scene)  # built by the host from the same two raw scans per index
    seg0 = [host.frontend_segment(host.synth_raw_scan(40 + i, 0, scene=scene)) for i in range(n)]
    seg1 = [host.frontend_segment(host.synth_raw_scan(40 + i, 1, scene=scene)) for i in range(n)]
    boot = np.zeros((n, 19))
    for i, p in enumerate(pairs):  # bootstrap pose = what the synthetic pairs re-project the first scan with
        boot[i, 0:3], boot[i, 6:10] = p.meta["true_t"], p.meta["true_q"]
    cov0 = np.tile(np.eye(18)[None] * 1e-4, (n, 1, 1))
    prm = pkg.default_params(num_iter=30)
    with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384) as c:
        want = c.update_batch(pairs)
        c.streams_init(n)
        r0, cnt0 = c.streams_step(seg0, boot, cov0)
        for i, r in enumerate(r0):  # first scan: nothing to match, state handed back
            assert r.iters == 0 and np.array_equal(r.state, boot[i])
        for i, p in enumerate(pairs):  # the resident clouds are the pair's targets (re-projected first-scan clouds)
            for which, ref in ((0, p.corner_last), (1, p.surf_last)):
                got = c.streams_peek(i, which)
                assert got.shape == ref.shape
                # bit-equal except where a time tag / a trig call differs in its last bit (device vs host libm)
                err = np.abs(got[:, :3].astype(np.float64) - ref[:, :3])
                # (one f32 step of a ring-15 time tag moves a point at 50 m by ~2e-5 m through the de-skew)
                assert err.max(initial=0) <= 5e-5 and (err > 0).mean() <= 2e-2 and np.abs(got[:, 3] - ref[:, 3]).max() <= 4e-6
        r1, cnt1 = c.streams_step(seg1, np.stack([p.state for p in pairs]), np.stack([p.cov for p in pairs]))
        fe_ms, up_ms, rp_ms = c.streams_stats()
        assert fe_ms > 0 and up_ms > 0 and rp_ms > 0
        for i, (g, w, p) in enumerate(zip(r1, want, pairs)):
            assert tuple(cnt1[i][[2, 0]]) == (len(p.surf_flat), len(p.corner_sharp))
            assert (g.iters, g.converged, g.diverged, g.m_surf, g.m_corner) == \
                   (w.iters, w.converged, w.diverged, w.m_surf, w.m_corner), i
            assert np.abs(g.state[:3] - w.state[:3]).max() <= 1e-6 and np.abs(g.state[6:10] - w.state[6:10]).max() <= 1e-7
            assert np.abs(g.cov - w.cov).max() <= 1e-6 * np.abs(w.cov).max()
        # the same two steps from RAW clouds (image projection on the device too): identical results, bit for bit
        c.streams_init(n)
        raw0 = [host.synth_raw_scan(40 + i, 0, scene=scene) for i in range(n)]
        raw1 = [host.synth_raw_scan(40 + i, 1, scene=scene) for i in range(n)]
        c.streams_step_raw(raw0, boot, cov0)
        r1r, cnt1r = c.streams_step_raw(raw1, np.stack([p.state for p in pairs]), np.stack([p.cov for p in pairs]))
        assert np.array_equal(cnt1r, cnt1)
        for a, b in zip(r1r, r1):
            assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov) and a.iters == b.iters
        # third step on the same context (slots swap back): still runs, poses stay finite
        r2, _ = c.streams_step(seg0, np.stack([p.state for p in pairs]), np.stack([p.cov for p in pairs]))
        assert all(np.isfinite(r.state).all() for r in r2)


def test_update_point_cloud_as_one_kernel_leaves_the_same_bits(pkg, ieskf, host, monkeypatch):
    """lins_streams_step re-projects a scan's clouds and builds their search index in ONE kernel (grid_index_kernel<true>);
    with the debug knob LINS_STREAMS_FUSE=0 the two run as the two kernels they were: same resident clouds, same states
    and covariances, bit for bit, over three steps (the slots swap twice)."""
    n = 5
    segs = [[host.frontend_segment(host.synth_raw_scan(60 + i, k)) for i in range(n)] for k in (0, 1)]
    pairs = host.synth_batch(n, start=60)
    boot = np.zeros((n, 19))
    for i, p in enumerate(pairs):
        boot[i, 0:3], boot[i, 6:10] = p.meta["true_t"], p.meta["true_q"]
    cov0 = np.tile(np.eye(18)[None] * 1e-4, (n, 1, 1))
    st = np.stack([p.state for p in pairs]); cv = np.stack([p.cov for p in pairs])

    def run(fuse):
        monkeypatch.setenv("LINS_ENABLE_DEBUG_KNOBS", "1")
        monkeypatch.setenv("LINS_STREAMS_FUSE", "1" if fuse else "0")
        out = []
        with ieskf.IeskfContext(pkg.default_params(num_iter=12), max_batch=n, max_targets=16384) as c:
            c.streams_init(n)
            c.streams_step(segs[0], boot, cov0)
            out.append([c.streams_peek(i, w) for i in range(n) for w in (0, 1)])
            for k in (1, 0):
                r, _ = c.streams_step(segs[k], st, cv)
                out.append([(x.state.copy(), x.cov.copy(), x.iters) for x in r])
                out.append([c.streams_peek(i, w) for i in range(n) for w in (0, 1)])
        return out

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            if isinstance(u, tuple):
                assert np.array_equal(u[0], v[0]) and np.array_equal(u[1], v[1]) and u[2] == v[2]
            else:
                assert np.array_equal(u, v)
    assert any(t[2] > 0 for t in a[1])  # (the second step really matched something)


def test_streams_step_in_several_parts_leaves_the_same_bits(pkg, ieskf, host, monkeypatch):
    """More streams than the device has workgroup slots (two per CU): the update of a step runs as several parts that hand
    the loop state over (the relay, as in lins_batch_run).  Same states, covariances and iteration counts as whole
    updates (LINS_RELAY_AT=0), bit for bit."""
    n_dist = 8
    n = 2 * 256 + 9  # (an MI355X has 256 CUs; an odd count, so that parts land on different XCDs)
    segs = [[host.frontend_segment(host.synth_raw_scan(70 + i, k)) for i in range(n_dist)] for k in (0, 1)]
    pairs = host.synth_batch(n_dist, start=70)
    tile = lambda xs: [xs[i % n_dist] for i in range(n)]
    boot = np.zeros((n, 19))
    for i in range(n):
        boot[i, 0:3], boot[i, 6:10] = pairs[i % n_dist].meta["true_t"], pairs[i % n_dist].meta["true_q"]
    cov0 = np.tile(np.eye(18)[None] * 1e-4, (n, 1, 1))
    st = np.stack([p.state for p in tile(pairs)]); cv = np.stack([p.cov for p in tile(pairs)])

    def run(relay_at):
        monkeypatch.setenv("LINS_ENABLE_DEBUG_KNOBS", "1")
        monkeypatch.setenv("LINS_RELAY_AT", str(relay_at))
        with ieskf.IeskfContext(pkg.default_params(num_iter=10, fixed_iters=1), max_batch=n, max_targets=16384) as c:
            c.streams_init(n)
            c.streams_step(tile(segs[0]), boot, cov0)
            r, _ = c.streams_step(tile(segs[1]), st, cv)
            return [(x.state.copy(), x.cov.copy(), x.iters) for x in r]

    a, b = run(4), run(0)
    assert all(t[2] == 10 for t in a)
    for u, v in zip(a, b):
        assert np.array_equal(u[0], v[0]) and np.array_equal(u[1], v[1]) and u[2] == v[2]
    for i in range(n_dist, n):  # (the same scan on another stream: the same bits)
        assert np.array_equal(a[i][0], a[i % n_dist][0])


def test_device_segmentation_matches_the_host_restatement(pkg, ieskf, host):
    """image_projection_node (IP:191-415) on the device == the host restatement, bit for bit: projection
    (last point owns a cell), ground flags, the BFS labelling restated as a min-label propagation over the
    reference's directed neighbour table (validity of every segment), emission order, ring indices,
    orientations, outlier count — and the feature stage fed from it yields the same feature clouds."""
    raws = [host.synth_raw_scan(idx, k) for idx in (0, 1, 2, 7) for k in (0, 1)]
    want = [host.frontend_segment(r) for r in raws]
    with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
        got = c.segment_batch(raws)
        assert c.segment_ms() > 0
        for g, w in zip(got, want):
            assert g.n == w.n and g.c.n_outlier == w.c.n_outlier
            assert list(g.c.start_ring) == list(w.c.start_ring) and list(g.c.end_ring) == list(w.c.end_ring)
            assert (g.c.start_ori, g.c.end_ori, g.c.ori_diff) == (w.c.start_ori, w.c.end_ori, w.c.ori_diff)
            n = w.n
            assert np.array_equal(g.cloud[:n], w.cloud[:n]) and np.array_equal(g.range[:n], w.range[:n])
            assert np.array_equal(g.col[:n], w.col[:n]) and np.array_equal(g.ground[:n], w.ground[:n])
        feats = c.extract_features_batch(got)
    for f, w in zip(feats, want):
        ref = host.frontend_extract_segmented(w)
        for k in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"):
            assert np.array_equal(f[k], ref[k]), k


def _wide_room_raw_scan(seed, r0=95.0, wobble=4.0):
    """A raw 16 x 1800 cloud in firing order (columns from +179.8 deg clockwise, the 16 rings of a column together)
    of a round hall of radius r0 + wobble cos(3 az) metres with a floor at z = -1.8: the beams meet the wall head-on,
    so every column of the upper rings survives the segmentation — a sector has 300 points (the 512-key sort
    network) — and the wall is far enough that a ring's VoxelGrid box exceeds 2^21 cells (the 64-bit voxel keys)."""
    rng = np.random.default_rng(seed)
    az = np.radians(179.83 - 0.2 * np.arange(1800))  # (off the column edges of IP:225)
    el = np.radians(-15.0 + 2.0 * np.arange(16))
    A, E = np.meshgrid(az, el, indexing="ij")  # (1800, 16): column-major firing order
    t_wall = (r0 + wobble * np.cos(3 * A)) / np.cos(E)
    with np.errstate(divide="ignore"):
        t_ground = np.where(E < 0, 1.8 / -np.sin(E), np.inf)
    t = np.minimum(t_wall, t_ground) + rng.normal(0, 0.02, A.shape)
    pts = np.stack([t * np.cos(E) * np.cos(A), t * np.cos(E) * np.sin(A), t * np.sin(E), np.zeros_like(t)], -1)
    return pts.reshape(-1, 4).astype(np.float32)


def test_front_end_on_a_wide_hall_takes_the_large_sort_paths(pkg, ieskf, host):
    """sectors of 300 points (512-key sector sort), rings of 1800 points whose voxel box needs 64-bit keys: the
    device stages still equal the host restatement bit for bit"""
    raws = [_wide_room_raw_scan(5), _wide_room_raw_scan(6, 60.0, 2.0)]
    want = [host.frontend_segment(r) for r in raws]
    assert max(int(e) - int(s) for w in want for s, e in zip(w.c.start_ring, w.c.end_ring)) > 1700
    with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
        got = c.segment_batch(raws)
        for g, w in zip(got, want):
            n = w.n
            assert g.n == n and g.c.n_outlier == w.c.n_outlier
            assert list(g.c.start_ring) == list(w.c.start_ring) and list(g.c.end_ring) == list(w.c.end_ring)
            assert np.array_equal(g.cloud[:n], w.cloud[:n]) and np.array_equal(g.range[:n], w.range[:n])
            assert np.array_equal(g.col[:n], w.col[:n]) and np.array_equal(g.ground[:n], w.ground[:n])
        feats = c.extract_features_batch(got)
    for f, w in zip(feats, want):
        ref = host.frontend_extract_segmented(w)
        assert len(ref["surf_less_flat"]) > 3000
        for k in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"):
            assert np.array_equal(f[k], ref[k]), k


def test_front_end_flip_in_the_last_partial_wave(pkg, ieskf, host):
    """undistortPcl's halfPassed flip (SE:631-638) found by the wave that straddles the end of the cloud: clouds of n points,
    n % 64 != 0, cut so that the flip index lies among the last n % 64 points — the device pass reduces the flip over the
    wave there, and every lane has to be present for it (the loop's trip count used to differ inside that wave).
    A wrong flip shows in every relative-time tag; the feature clouds still equal the host restatement bit for bit."""
    segs, flips = [], []
    for seed in (21, 22, 23):
        w = host.frontend_segment(host.synth_raw_scan(seed, 0))
        x, y = w.cloud[:w.n, 0].astype(np.float64), w.cloud[:w.n, 1].astype(np.float64)
        s_ori = float(w.c.start_ori)
        ori = -np.arctan2(y, x)
        ori = np.where(ori < s_ori - np.pi / 2, ori + 2 * np.pi, np.where(ori > s_ori + 1.5 * np.pi, ori - 2 * np.pi, ori))
        flip = int(np.argmax(ori - s_ori > np.pi))  # (ring 0 sweeps the whole turn: somewhere in its second half)
        assert 64 < flip < w.n - 70
        n = flip + 3 + (1 if (flip + 3) % 64 == 0 else 0)  # the flip among the last n % 64 points of the cut cloud
        assert n % 64 != 0 and (n - 1) // 64 == flip // 64 and n - (n % 64) <= flip < n
        # (the cut cloud is one ring — ring 0 up to index n: the other rings are empty, SE:731-735's sp / ep from IP:296, 320)
        start = [n - 1 + 5] * 16
        end = [n - 1 - 5] * 16
        start[0], end[0] = int(w.c.start_ring[0]), n - 1 - 5
        assert int(w.c.end_ring[0]) + 5 >= n - 1  # (the cut lies inside ring 0)
        segs.append(host.segmented_from_arrays(w.cloud[:n], w.range[:n], w.col[:n], w.ground[:n], n, start, end,
                                               (w.c.start_ori, w.c.end_ori, w.c.ori_diff), 0))
        flips.append((flip, n))
    with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
        feats = c.extract_features_batch(segs)
    for (flip, n), f, w in zip(flips, feats, segs):
        ref = host.frontend_extract_segmented(w)
        assert len(ref["surf_less_flat"]) > 20, (flip, n)
        for k in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"):
            assert np.array_equal(f[k], ref[k]), (k, flip, n)


def test_front_end_on_rough_ranges_takes_the_many_candidates_paths(pkg, ieskf, host):
    """Round 4's front-end picks the edge candidates of a sector one per lane when there are at most 64 of them and from
    per-lane bit masks otherwise, and the plane candidates always from the masks; ties between equal curvatures go by index.
    Scans whose ranges are rough (ripples of 0.3 m: hundreds of points of curvature > 0.5 per sector, many of them ground), quantised (ranges rounded to 0.25 m: equal curvatures everywhere) and with ground everywhere
    (every sector has plane candidates in all its 64-element blocks) still give the host restatement's feature clouds, bit
    for bit."""
    base = [host.frontend_segment(host.synth_raw_scan(90 + i, i % 2)) for i in range(3)]
    rng = np.random.default_rng(5)
    segs = []
    for k, w in enumerate(base):
        n = w.n
        r = w.range[:n].copy(); g = w.ground[:n].copy()
        if k == 0:
            # (smooth ripples: curvature up to 8 without the range jumps of 0.3 m that mark a neighbourhood as occluded, SE:691-703)
            r += (0.3 * np.sin(0.5 * np.arange(n) + rng.uniform(0, 6.28))).astype(np.float32)
        elif k == 1:
            r = (np.round(r * 4) / 4).astype(np.float32)
        else:
            g[:] = 1
        d = np.zeros(n)
        for o in range(-5, 6):
            d[5:n - 5] += (r[5 + o:n - 5 + o] if o else -10 * r[5:n - 5])
        if k == 0:
            per_sector = [int(((d[s + (e - s) * j // 6:s + (e - s) * (j + 1) // 6] ** 2) > 0.5).sum())
                          for s, e in zip(w.c.start_ring, w.c.end_ring) for j in range(6) if e > s]
            assert max(per_sector) > 128  # (beyond one candidate per lane: the mask path)
        segs.append(host.segmented_from_arrays(w.cloud[:n], r, w.col[:n], g, n, list(w.c.start_ring), list(w.c.end_ring),
                                               (w.c.start_ori, w.c.end_ori, w.c.ori_diff), w.c.n_outlier))
    with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
        feats = c.extract_features_batch(segs)
    for i, (f, w) in enumerate(zip(feats, segs)):
        ref = host.frontend_extract_segmented(w)
        # (with ground everywhere nothing is an edge, SE:753, and every sector yields its four planes)
        counts = {k: len(v) for k, v in ref.items() if hasattr(v, "shape")}
        assert (counts["corner_less_sharp"] > 50) if i < 2 else (counts["corner_less_sharp"] == 0 and counts["surf_flat"] > 50), (i, counts)
        for k in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"):
            assert np.array_equal(f[k], ref[k]), (i, k)


def test_segmentation_of_a_cloud_with_more_points_than_cells(pkg, ieskf, host):
    """more raw points than the 28 800 cells (a driver that repeats packets): later points take over their cells
    (IP:238-240), and the kernel's cell-by-cell path (clouds beyond 32 768 points) equals the host restatement"""
    base = host.synth_raw_scan(3, 1)
    rng = np.random.default_rng(11)
    extra = base[rng.permutation(len(base))[:14000]].copy()
    extra[:, :3] *= np.float32(1.01)  # the same directions, a little farther: other ranges in the same cells
    raw = np.ascontiguousarray(np.concatenate([base, extra]))
    assert len(raw) > 32768
    want = host.frontend_segment(raw)
    with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
        g = c.segment_batch([raw, base])[0]
    n = want.n
    assert g.n == n and g.c.n_outlier == want.c.n_outlier
    assert list(g.c.start_ring) == list(want.c.start_ring) and list(g.c.end_ring) == list(want.c.end_ring)
    assert np.array_equal(g.cloud[:n], want.cloud[:n]) and np.array_equal(g.range[:n], want.range[:n])
    assert np.array_equal(g.col[:n], want.col[:n]) and np.array_equal(g.ground[:n], want.ground[:n])


def test_streams_step_keeps_the_healthy_streams_when_one_cannot_take_the_fallback(pkg, ieskf, host):
    """ICP_FREQ = 2: the device ICP fallback is not available, so a diverged stream (NaN prior covariance ->
    diverged = 2, SE:552-563) keeps its un-updated filter and is flagged in ITS result — the step still completes for
    every stream (slots flipped, clouds re-projected) and the next step runs."""
    n = 3
    seg0 = [host.frontend_segment(host.synth_raw_scan(60 + i, 0)) for i in range(n)]
    seg1 = [host.frontend_segment(host.synth_raw_scan(60 + i, 1)) for i in range(n)]
    pairs = host.synth_batch(n, start=60)
    boot = np.zeros((n, 19))
    for i, p in enumerate(pairs):
        boot[i, 0:3], boot[i, 6:10] = p.meta["true_t"], p.meta["true_q"]
    cov = np.stack([p.cov for p in pairs])
    prm = pkg.default_params(num_iter=12, icp_freq=2)
    with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384) as c:
        c.streams_init(n)
        c.streams_step(seg0, boot, cov)
        bad_cov = cov.copy()
        bad_cov[1] = np.nan
        prior = np.stack([p.state for p in pairs])
        r1, _ = c.streams_step(seg1, prior, bad_cov)
        assert r1[1].diverged == 2 and r1[1].reserved[0] == -7  # LINS_E_UNSUPPORTED, this stream only
        assert np.array_equal(r1[1].state, prior[1])
        for k in (0, 2):
            assert r1[k].diverged == 0 and r1[k].reserved[0] == 0 and r1[k].iters > 0
            assert np.abs(r1[k].state[:3] - pairs[k].meta["true_t"]).max() < 0.1
        r2, _ = c.streams_step(seg0, prior, cov)  # the context is still consistent: slots swapped back, everybody steps
        assert all(np.isfinite(r.state).all() and r.iters > 0 for r in r2)


def test_pipelined_batch_call_equals_the_staged_path(pkg, oracle, ieskf):
    """lins_ieskf_update_batch pipelines large batches in chunks (pack || H2D || kernels): same bits as
    upload / run / download of the whole batch; a chunk with an oversized scan takes the global-memory grid
    on its own and still matches the oracle; a contract violation in a later chunk fails the whole call."""
    rng = np.random.default_rng(77)
    base = [make_pair(pkg, rng, 40 + 7 * k, 50 + 5 * k, 600 + 90 * k, 200 + 30 * k, "offgrid") for k in range(7)]
    base.append(make_pair(pkg, rng, 0, 0, 500, 100, "offgrid"))  # (no queries)
    n = 1100 + 37
    pairs = [base[(k * 5 + k // 8) % len(base)] for k in range(n)]
    c = ieskf.IeskfContext(pkg.default_params(num_iter=30), device=0, max_batch=n, max_targets=13000)
    try:
        for search in ("auto", "mr", "lds"):
            c.set_search(search)
            got = c.update_batch(pairs)
            c.run()  # the batch of the call stays resident (inputs, search index, launch order): a staged run on it
            c.sync()
            rerun = c.download(len(pairs))
            c.upload(pairs)
            c.run()
            want = c.download()
            for k, (g, w) in enumerate(zip(rerun, want)):
                assert np.array_equal(g.state.view(np.int64), w.state.view(np.int64)), (search, k)
                assert np.array_equal(g.cov.view(np.int64), w.cov.view(np.int64)), (search, k)
            for k, (g, w) in enumerate(zip(got, want)):
                assert (g.iters, g.converged, g.diverged, g.m_surf, g.m_corner) == (w.iters, w.converged, w.diverged, w.m_surf, w.m_corner), (search, k)
                assert np.array_equal(g.state.view(np.int64), w.state.view(np.int64)), (search, k)
                assert np.array_equal(g.cov.view(np.int64), w.cov.view(np.int64)), (search, k)
        c.set_search("auto")
        mixed = list(pairs)
        mixed[900] = make_pair(pkg, rng, 40, 50, 12000, 900, "offgrid")  # does not fit LDS: its chunk falls back
        got = c.update_batch(mixed)
        for k in (0, 511, 512, 899, 900, 901, n - 1):
            want = oracle.ieskf(pkg.default_params(num_iter=30), mixed[k], oracle.FORM_DENSE, oracle.NN_BRUTE)
            g = got[k]
            assert (g.iters, g.converged, g.diverged, g.m_surf, g.m_corner) == (want.iters, want.converged, want.diverged, want.m_surf, want.m_corner), k
            if not want.diverged:
                assert np.abs(g.state - want.state).max() <= 1e-6 * max(1.0, np.abs(want.state).max()), k
        bad = list(pairs)
        p = bad[1000]
        sf = p.surf_flat.copy()
        sf[3, 0] = np.nan
        bad[1000] = pkg.ScanPair(sf, p.corner_sharp, p.surf_last, p.corner_last, p.state, p.cov)
        with pytest.raises(Exception):
            c.update_batch(bad)
        again = c.update_batch(pairs)  # the context is usable after the failed call
        assert all(np.isfinite(r.state).all() for r in again[:16])
    finally:
        c.close()


def test_front_end_centroids_of_voxels_that_span_many_chunks(pkg, ieskf, host):
    """Round 5's centroid pass carries a voxel's partial sums from one 64-position chunk of the sorted order into the next
    and finishes a run alone only at the end of a wave's group of chunks.  Clouds shrunk towards the origin put hundreds
    (x 0.02) or all (x 0.001: one or two voxels per ring, runs of ~1700 points across every chunk and group border) of a
    ring's kept points into one 0.2 m voxel; picks are untouched (they follow the ranges).  Same f32 sums, in the same
    order, as the host restatement: bit for bit."""
    base = [host.frontend_segment(host.synth_raw_scan(40 + i, i % 2)) for i in range(2)]
    segs = []
    for w in base:
        n = w.n
        for scale in (0.02, 0.001):
            cloud = w.cloud[:n].copy()
            cloud[:, :3] *= np.float32(scale)
            segs.append(host.segmented_from_arrays(cloud, w.range[:n], w.col[:n], w.ground[:n], n, list(w.c.start_ring), list(w.c.end_ring),
                                                   (w.c.start_ori, w.c.end_ori, w.c.ori_diff), w.c.n_outlier))
    with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
        feats = c.extract_features_batch(segs)
    for i, (f, w) in enumerate(zip(feats, segs)):
        ref = host.frontend_extract_segmented(w)
        if i % 2:  # (everything within a few centimetres of the origin: a handful of voxels for the whole scan)
            assert len(ref["surf_less_flat"]) <= 16 * 8, len(ref["surf_less_flat"])
        for k in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"):
            assert np.array_equal(f[k], ref[k]), (i, k)


def test_front_end_on_empty_and_tiny_segmented_scans(pkg, ieskf, host):
    """A segmented scan without a single point (first, in the middle and last of a batch) and scans of a handful of
    points: no feature, no fault, the neighbours' results untouched — and the host restatement says the same."""
    full = [host.frontend_segment(host.synth_raw_scan(9, k)) for k in (0, 1)]

    def cut(seg, n):
        c = host.segmented_from_arrays(seg.cloud[:max(n, 1)].copy(), seg.range[:max(n, 1)].copy(), seg.col[:max(n, 1)].copy(),
                                       seg.ground[:max(n, 1)].copy(), n, [-1 + 5] * 16 if n == 0 else list(np.minimum(seg.c.start_ring, n)),
                                       [-1 - 5] * 16 if n == 0 else list(np.minimum(seg.c.end_ring, n - 1)),
                                       (seg.c.start_ori, seg.c.end_ori, seg.c.ori_diff), 0)
        return c

    segs = [cut(full[0], 0), full[0], cut(full[1], 0), cut(full[0], 7), full[1], cut(full[1], 40), cut(full[0], 0)]
    want = [host.frontend_extract_segmented(s) for s in segs]
    with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
        got = c.extract_features_batch(segs)
    for i, (g, w) in enumerate(zip(got, want)):
        for k in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"):
            assert g[k].shape == w[k].shape and np.array_equal(g[k], w[k]), (i, k)
    assert all(len(got[i][k]) == 0 for i in (0, 2, 6) for k in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"))


@pytest.mark.parametrize("search", ["mr", "lds", "binned"])
def test_non_finite_prior_states_take_the_nan_branch_like_the_oracle(pkg, ieskf, host, oracle, search):
    """A prior state with a NaN or an infinity in it (position, quaternion, velocity, bias) is not an input error in the
    reference: the first iteration produces NaN, SE:552-563 flags it (diverged = 2) and the filter keeps its state.  The
    device does the same — same flags and row counts as the oracle, no hang — next to a healthy scan of the same batch."""
    prm = pkg.default_params(num_iter=10)
    base = [host.synth_pair(i) for i in range(4)]
    cases = []
    for k, (idx, val) in enumerate([(0, np.nan), (7, np.nan), (3, np.inf), (12, np.nan)]):
        p = base[k]
        st = p.state.copy()
        st[idx] = val
        cases.append(pkg.ScanPair(p.surf_flat, p.corner_sharp, p.surf_last, p.corner_last, st, p.cov))
    cases.append(base[0])
    with ieskf.IeskfContext(prm, max_batch=len(cases), max_targets=16384, search=search) as c:
        got = c.update_batch(cases)
    for g, p in zip(got, cases):
        w = oracle.ieskf(prm, p, oracle.FORM_DENSE, oracle.NN_KDTREE)
        assert (g.iters, g.converged, g.diverged, g.m_surf, g.m_corner) == (w.iters, w.converged, w.diverged, w.m_surf, w.m_corner)
        assert np.array_equal(np.isnan(g.state), np.isnan(w.state))
    assert [g.diverged for g in got] == [2, 2, 2, 2, 0]
