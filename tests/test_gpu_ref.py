"""The HIP path, through the C ABI, against the REFERENCE'S OWN SOURCES (oracle/_ref/liblins_ref.so =
/root/reference/lins/include/StateEstimator.hpp compiled verbatim against stand-in third-party headers; it is
built in the container that has /root/reference and travels to the GPU box with the snapshot), and the batch
bench.py times — all 1024 scans of it — against the oracle's faithful dense form.

Bars (BASELINE.json north_star / SURVEY.md §8d): index triplets and accepted sets bit-exact; f32 rows and
de-skewed points bit-exact up to <= 1 ulp on <= 1e-3 of the values (ocml vs glibc sin / cos / atan2 in the last f64
ulp before the cast); flags and iteration counts equal; |dp| <= 1e-6 m, |dq| <= 1e-7, max|dP| <= 1e-9 max|P|.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from test_gpu_parity import assert_corr_equal, assert_result_close
from test_ref import widen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    from oracle import ref as r

    if not r.available():
        if os.environ.get("LINS_REQUIRE_REF") == "1":  # (the default on a GPU box, tests/conftest.py: a lost checker must not read as "green")
            pytest.fail("LINS_REQUIRE_REF=1 and oracle/_ref/liblins_ref.so did not travel with the snapshot")
        pytest.skip("oracle/_ref/liblins_ref.so did not travel and /root/reference is not here to build it")
    r.lib()
    return r


def cores():
    return min(64, os.cpu_count() or 1)


@pytest.mark.parametrize("search", ["auto", "mr", "lds", "lds1", "binned", "brute"])
def test_correspondences_bit_exact_along_the_references_trajectory(pkg, ieskf, host, ref, search):
    """Device findCorresponding*Features at every linearisation state the REFERENCE visits (its own performIESKF,
    replayed with NUM_ITER = 1, 2, ...), against the reference's own pointSearch*Ind / coefficient rows."""
    prm = pkg.default_params(num_iter=30)
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384, search=search) as ctx:
        for idx in (0, 3, 41):
            pair = host.synth_pair(idx)
            states = [pair.state] + [r.state for r, _ in ref.replay(prm, pair)]
            for k, lin in enumerate(states[:-1]):
                want_s, want_c = ref.correspondences(prm, pair, lin, k)
                surf, corner = ctx.correspondences(pair, lin, k)
                assert_corr_equal(surf, want_s, f"pair{idx}.iter{k}.surf")
                assert_corr_equal(corner, want_c, f"pair{idx}.iter{k}.corner")


@pytest.mark.parametrize("search", ["auto", "mr", "lds", "lds1", "binned", "brute"])
@pytest.mark.parametrize("wide", [False, True], ids=["shipped-prior", "wide-prior"])
def test_update_matches_the_references_perform_ieskf(pkg, ieskf, host, ref, search, wide):
    """performIESKF (reference stop rule, NUM_ITER 30) on 96 seeded pairs, as one batch: HIP vs the reference's code."""
    prm = pkg.default_params(num_iter=30)
    start = 3000 if not wide else 8000
    pairs = host.synth_batch(96, start=start)
    if wide:
        widen(pairs, start)
    want = ref.perform_ieskf_batch(prm, pairs, threads=cores())
    with ieskf.IeskfContext(prm, max_batch=len(pairs), max_targets=16384, search=search) as ctx:
        got = ctx.update_batch(pairs)
    # (the contract's bars instead of the tight ones only for the updates the REFERENCE itself ran to NUM_ITER un-converged —
    # counted, not guessed: see assert_result_close)
    budget = [sum(1 for w in want if w.iters >= 30 and not w.converged and not w.diverged)]
    for g, w in zip(got, want):
        assert_result_close(g, w, budget)


@pytest.mark.parametrize("search", ["auto", "mr", "lds", "lds1", "binned", "brute"])
def test_open_scene_family_matches_the_reference(pkg, ieskf, host, ref, search):
    """The second scene family (open ground, ~60 trunks, far wall segments, 30 % of the returns lost, a moving box:
    csrc/host/synth.cpp) through every search mode: correspondences along the reference's trajectory bit for bit,
    performIESKF on 96 pairs as one batch against the reference's own code.  Sparse clouds (~2 k targets), many empty
    grid cells, queries with nothing inside the search radius — what the room never showed the grid, the certificates'
    margins and the walk cache."""
    prm = pkg.default_params(num_iter=30)
    start = 41000
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384, search=search) as ctx:
        for idx in (start, start + 5):
            pair = host.synth_pair(idx, scene=1)
            states = [pair.state] + [r.state for r, _ in ref.replay(prm, pair)]
            for k, lin in enumerate(states[:-1]):
                want_s, want_c = ref.correspondences(prm, pair, lin, k)
                surf, corner = ctx.correspondences(pair, lin, k)
                assert_corr_equal(surf, want_s, f"open pair{idx}.iter{k}.surf")
                assert_corr_equal(corner, want_c, f"open pair{idx}.iter{k}.corner")
    pairs = host.synth_batch(96, start=start + 100, scene=1)
    want = ref.perform_ieskf_batch(prm, pairs, threads=cores())
    with ieskf.IeskfContext(prm, max_batch=len(pairs), max_targets=16384, search=search) as ctx:
        got = ctx.update_batch(pairs)
    for g, w in zip(got, want):
        assert_result_close(g, w)


def test_open_scene_family_in_several_parts_and_both_input_layouts(pkg, ieskf, host, defs):
    """600 open-scene pairs on the batch kernel: several-part updates (tickets), pcl::PointXYZI-strided inputs and the
    mapped staging arena return the whole updates' bits from packed inputs."""
    prm = pkg.default_params(num_iter=30)
    pairs = host.synth_batch(600, start=42000, scene=1)
    with ieskf.IeskfContext(prm, max_batch=len(pairs), max_targets=16384, search="mr") as c:
        c.upload(pairs[:300])  # (within the device's workgroup slots: whole updates)
        c.run(); c.sync()
        whole = c.download()
        assert c.last_cut()[0] == 1
        c.upload(pairs)
        c.run(); c.sync()
        cut = c.download()
        assert c.last_cut()[0] > 1
        arr32, keep = defs.pairs_strided(pairs)
        got32 = c.update_batch(pairs, arr=arr32)
        gotm = c.update_batch(pairs, arr=c.map_batch(pairs))
    for a, b in zip(whole, cut[:300]):
        assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov) and a.iters == b.iters
    for a, b, m in zip(cut, got32, gotm):
        assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov) and a.iters == b.iters
        assert np.array_equal(a.state, m.state) and np.array_equal(a.cov, m.cov)


def test_divergence_and_fallback_match_the_reference(pkg, ieskf, ref):
    """SE:566-570 -> SE:585-592 through lins_host_perform_ieskf (device loop, device ICP) vs the reference."""
    from diverging import make_diverging_pair

    prm = pkg.default_params(num_iter=30)
    pair = make_diverging_pair(pkg)
    want = ref.perform_ieskf(prm, pair)
    assert want.diverged == 1
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384) as ctx:
        raw = ctx.update(pair)
        assert (raw.diverged, raw.iters) == (want.diverged, want.iters)
        got, used_fallback = ctx.perform_ieskf(pair)
        assert used_fallback
    assert np.abs(got.state[:3] - want.state[:3]).max() <= 1e-6 and np.abs(got.state[6:10] - want.state[6:10]).max() <= 1e-7
    assert np.abs(got.cov - want.cov).max() <= 1e-9 * max(1.0, np.abs(want.cov).max())


def test_bench_batch_all_1024_scans_against_the_dense_oracle_and_the_reference(pkg, ieskf, host, oracle, ref):
    """BASELINE.json configs[3], exactly what bench.py times: host.synth_pair(0..1023), search "auto" (=> the
    multi-resident kernel), 10 fixed iterations — every scan against the oracle's faithful dense M x M form with
    kd-tree neighbours (SE:542-549); and the same batch under the reference's stop rule against the reference's
    own code (which has no fixed-iteration mode)."""
    n = 1024
    with ThreadPoolExecutor(cores()) as ex:
        pairs = list(ex.map(host.synth_pair, range(n)))
    fixed = pkg.default_params(num_iter=10, fixed_iters=1)
    with ieskf.IeskfContext(fixed, max_batch=n, max_targets=16384, search="auto") as ctx:
        ctx.upload(pairs)  # the staged path bench.py uses
        ctx.run()
        ctx.sync()
        got = ctx.download()
    with ThreadPoolExecutor(cores()) as ex:
        want = list(ex.map(lambda p: oracle.ieskf(fixed, p, oracle.FORM_DENSE, oracle.NN_KDTREE), pairs))
    assert sum(w.iters for w in want) == 10 * n
    for g, w in zip(got, want):
        assert_result_close(g, w)
    stop = pkg.default_params(num_iter=30)
    want = ref.perform_ieskf_batch(stop, pairs, threads=cores())
    with ieskf.IeskfContext(stop, max_batch=n, max_targets=16384, search="auto") as ctx:
        got = ctx.update_batch(pairs)
    budget = [sum(1 for w in want if w.iters >= 30 and not w.converged and not w.diverged)]  # (the updates the reference ran out on)
    for g, w in zip(got, want):
        assert_result_close(g, w, budget)


def test_device_segmentation_and_front_end_against_the_references_two_nodes(pkg, ieskf, host, ref):
    """Raw clouds through the device chain (lins_segment_batch -> lins_extract_features_batch) against the reference's
    own image_projection_node.cpp and StateEstimator feature stage (both compiled verbatim, oracle/_ref): the segmented
    cloud, ranges, columns, ground flags, ring indices and outlier count bit for bit (orientations: libm's atan2f vs the
    product's fixed sequence, <= 4 ulp); the feature clouds with the same picks and voxels (coordinates bit-equal where
    the order is, centroids to a few f32 ulps; time tags <= 2.5e-7 relative)."""
    from test_ref import assert_same_picks, assert_same_segmentation

    prm = pkg.default_params()
    raws = [host.synth_raw_scan(400 + i // 2, i & 1) for i in range(24)]
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=1024) as c:
        got = c.segment_batch(raws)
        feats = c.extract_features_batch(got)
    with ThreadPoolExecutor(cores()) as ex:
        want = list(ex.map(ref.segment, raws))
    for i, (g, r) in enumerate(zip(got, want)):
        assert_same_segmentation(r, g, f"scan {i}")
    for i in (0, 5, 11, 18):
        r = want[i]
        k = r.n
        seg = dict(cloud=r.cloud[:k], range=r.range[:k], col=r.col[:k], ground=r.ground[:k], n=k, start_ring=list(r.c.start_ring),
                   end_ring=list(r.c.end_ring), orientation=(r.c.start_ori, r.c.end_ori, r.c.ori_diff), n_outlier=r.c.n_outlier)
        fr = ref.extract_features(prm, seg)
        fd = feats[i]
        for name in ("corner_sharp", "corner_less_sharp", "surf_flat"):
            assert_same_picks(fr[name][:, :3], fd[name][:, :3], seg, fr["undistorted"][:, :3], name)
        a, b = fr["surf_less_flat"], fd["surf_less_flat"]
        assert a.shape == b.shape and np.abs(a - b).max() <= 4e-6


def test_open_scene_family_through_the_device_front_end(pkg, ieskf, host, ref):
    """The stages in front of the update on the second scene family (open ground to the range limit, trunks, far wall
    segments, 30 % of the returns lost, a moving box): sparse rings, sky, ragged segments — raw clouds through the device
    projection / segmentation and feature front-end against the reference's two nodes (segmented cloud bit for bit, same
    picks and voxels) and against the host restatement bit for bit on 64 scans."""
    from test_ref import assert_same_picks, assert_same_segmentation

    prm = pkg.default_params()
    raws = [host.synth_raw_scan(43000 + i // 2, i & 1, scene=1) for i in range(64)]
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=1024) as c:
        got = c.segment_batch(raws)
        feats = c.extract_features_batch(got)
    with ThreadPoolExecutor(cores()) as ex:
        want = list(ex.map(ref.segment, raws[:12]))
        hseg = list(ex.map(host.frontend_segment, raws))
        hfe = list(ex.map(host.frontend_extract_segmented, hseg))
    for i, (g, r) in enumerate(zip(got, want)):
        assert_same_segmentation(r, g, f"open scan {i}")
    for i in (0, 3, 7, 10):
        r = want[i]
        k = r.n
        seg = dict(cloud=r.cloud[:k], range=r.range[:k], col=r.col[:k], ground=r.ground[:k], n=k, start_ring=list(r.c.start_ring),
                   end_ring=list(r.c.end_ring), orientation=(r.c.start_ori, r.c.end_ori, r.c.ori_diff), n_outlier=r.c.n_outlier)
        fr = ref.extract_features(prm, seg)
        fd = feats[i]
        for name in ("corner_sharp", "corner_less_sharp", "surf_flat"):
            assert_same_picks(fr[name][:, :3], fd[name][:, :3], seg, fr["undistorted"][:, :3], name)
        a, b = fr["surf_less_flat"], fd["surf_less_flat"]
        assert a.shape == b.shape and np.abs(a - b).max() <= 4e-6
    for i, (g, w, f, r) in enumerate(zip(got, hseg, feats, hfe)):
        k = w.n
        assert g.n == k and np.array_equal(g.cloud[:k], w.cloud[:k]) and np.array_equal(g.range[:k], w.range[:k]), i
        assert np.array_equal(g.col[:k], w.col[:k]) and np.array_equal(g.ground[:k], w.ground[:k]), i
        for key in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"):
            assert np.array_equal(f[key], r[key]), (i, key)


def test_device_scan_to_map_against_the_references_mapping_node(pkg, ieskf, ref):
    """lins_map_correspondences / lins_scan2map_batch against the reference's own lidar_mapping_node.cpp (compiled verbatim,
    oracle/ref_map_driver.cpp): the rows cornerOptimization / surfOptimization push — same queries, coefficients bit for
    bit — and scan2MapOptimization's rounds, flags, selected rows, transform (<= 2e-5: the device sums the normal equations
    in an f64 tree, the node's restated GEMM in row order)."""
    import importlib

    from map_synth import make_corridor, make_problem

    defs = importlib.import_module("lins---lidar-inertial-slam_amd._ctypes_defs")
    probs = [make_problem(defs, 500 + s)[0] for s in range(6)] + [make_corridor(defs, 45)[0]]
    with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
        for k, p in enumerate(probs):
            ro, rc = ref.map_rows(p)
            gc, gs = c.map_correspondences(p)
            go = np.concatenate([p.scan_corner[gc["accepted"] == 1], p.scan_surf[gs["accepted"] == 1]])
            gco = np.concatenate([gc["coeff"][gc["accepted"] == 1], gs["coeff"][gs["accepted"] == 1]])
            assert ro.shape == go.shape and np.array_equal(ro.view(np.int32), go.view(np.int32)), k
            assert np.array_equal(rc.view(np.int32), gco.view(np.int32)), k
        got = c.scan2map_batch(probs)
    for k, (p, g) in enumerate(zip(probs, got)):
        w = ref.scan2map(p)
        assert (g["iters"], g["converged"], g["degenerate"], g["n_sel"]) == (w["iters"], w["converged"], w["degenerate"], w["n_sel"]), k
        assert np.abs(g["transform"] - w["transform"]).max() <= 2e-5, k
    assert got[-1]["degenerate"] == 1
