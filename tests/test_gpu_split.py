"""The split path ("split" search mode: grid kernel for the first iterations, list kernel for the rest,
csrc/ieskf_split.h): same results as the persistent "mr" kernel on a batch, the list kernel's index triplets equal
to the oracle's at the iterations it decides, its exhaustive-search fallback exercised and still exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def batch(host):
    return host.synth_batch(48)


@pytest.mark.parametrize("fixed", [1, 0])
def test_split_equals_the_persistent_kernel_on_a_batch(pkg, ieskf, batch, fixed):
    prm = pkg.default_params(num_iter=10 if fixed else 30, fixed_iters=fixed)
    res = {}
    for mode in ("mr", "split"):
        with ieskf.IeskfContext(prm, max_batch=len(batch), max_targets=16384, search=mode) as c:
            res[mode] = c.update_batch(batch)
            if mode == "split":
                grid_ms, list_ms = c.last_split_ms()
                assert grid_ms > 0 and list_ms > 0
    for a, b in zip(res["mr"], res["split"]):
        assert (a.iters, a.converged, a.diverged, a.m_surf, a.m_corner) == (b.iters, b.converged, b.diverged, b.m_surf, b.m_corner)
        # same rows, same 28 sums up to the order of the wave-level additions (different lane layout)
        assert np.abs(a.state - b.state).max() <= 1e-12
        assert np.abs(a.cov - b.cov).max() <= 1e-12 * np.abs(a.cov).max()


@pytest.mark.parametrize("it", [3, 4, 7, 9])
def test_list_kernel_triplets_equal_the_oracle_trace(pkg, ieskf, oracle, batch, it):
    prm = pkg.default_params(num_iter=10, fixed_iters=1)
    pairs = batch[:6]
    with ieskf.IeskfContext(prm, max_batch=len(pairs), max_targets=16384, search="split") as c:
        c.upload(pairs)
        c.split_dump_arm(it)
        c.run()
        c.sync()
        d = c.split_dump_read(sum(len(p.surf_flat) + len(p.corner_sharp) for p in pairs))
    off = 0
    for p in pairs:
        _, tr = oracle.ieskf(prm, p, oracle.FORM_DENSE, oracle.NN_BRUTE, trace=True)
        for kind, nq in (("surf", len(p.surf_flat)), ("corner", len(p.corner_sharp))):
            g, o = d[off:off + nq], tr[kind][it]
            off += nq
            assert np.array_equal(g["ind1"], o["ind1"]) and np.array_equal(g["ind2"], o["ind2"]), (kind, it)
            if kind == "surf":
                assert np.array_equal(g["ind3"], o["ind3"]), (kind, it)
            assert np.array_equal(g["accepted"] & 1, o["accepted"]), (kind, it)
            ulp = np.abs(g["coeff"].view(np.int32).astype(np.int64) - o["coeff"].view(np.int32).astype(np.int64))
            assert ulp.max(initial=0) <= 1 and (ulp > 0).mean() <= 1e-3


def test_uncertified_decisions_fall_back_to_the_exhaustive_walk_and_stay_exact(pkg, ieskf, batch, monkeypatch):
    """A 1 cm margin (debug knob) leaves most decisions uncertified after the first drift: the exhaustive searches
    — nearest neighbour over the whole cloud, literal index walk — then carry the result, which must not change."""
    prm = pkg.default_params(num_iter=10, fixed_iters=1)
    pairs = batch[:8]
    with ieskf.IeskfContext(prm, max_batch=len(pairs), max_targets=16384, search="mr") as c:
        want = c.update_batch(pairs)
    monkeypatch.setenv("LINS_ENABLE_DEBUG_KNOBS", "1")
    monkeypatch.setenv("LINS_SPLIT_MARGIN", "0.01")
    monkeypatch.setenv("LINS_SPLIT_ITERS", "2")
    with ieskf.IeskfContext(prm, max_batch=len(pairs), max_targets=16384, search="split") as c:
        got = c.update_batch(pairs)
    assert sum(r.reserved[2] for r in got) > 50 * len(pairs)  # (the fallback really ran, many times)
    for a, b in zip(want, got):
        assert (a.iters, a.converged, a.diverged, a.m_surf, a.m_corner) == (b.iters, b.converged, b.diverged, b.m_surf, b.m_corner)
        assert np.abs(a.state - b.state).max() <= 1e-12 and np.abs(a.cov - b.cov).max() <= 1e-12 * np.abs(a.cov).max()


def test_split_needs_icp_freq_1_else_runs_the_persistent_kernel(pkg, ieskf, oracle, batch):
    prm = pkg.default_params(num_iter=12, icp_freq=3)
    with ieskf.IeskfContext(prm, max_batch=2, max_targets=16384, search="split") as c:
        got = c.update_batch(batch[:2])
        with pytest.raises(ieskf.LinsError):
            c.last_split_ms()  # (the last run did not take the split path)
    for g, p in zip(got, batch[:2]):
        w = oracle.ieskf(prm, p, oracle.FORM_DENSE, oracle.NN_BRUTE)
        assert (g.iters, g.converged, g.diverged) == (w.iters, w.converged, w.diverged)
        assert np.abs(g.state[:3] - w.state[:3]).max() <= 1e-6


def test_handed_over_state_is_the_persistent_kernels(pkg, ieskf, host):
    """The grid kernel hands the list kernel the linearisation state after `split_iters` iterations: bit for bit the
    state the persistent kernel has at that point (lins_debug_split_hand)."""
    import ctypes as C

    pair = host.synth_pair(5)
    with ieskf.IeskfContext(pkg.default_params(num_iter=3, fixed_iters=1), max_batch=1, max_targets=16384, search="mr") as c:
        x3 = c.update(pair).state
    with ieskf.IeskfContext(pkg.default_params(num_iter=6, fixed_iters=1), max_batch=1, max_targets=16384, search="split") as c:
        c.upload([pair])
        c.run()
        c.sync()
        L = ieskf.lib()
        L.lins_debug_split_hand.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lin = np.zeros(19)
        st = np.zeros(2, np.int32)
        assert L.lins_debug_split_hand(c._h, 0, lin.ctypes.data, st.ctypes.data) == 0
    assert tuple(st) == (3, 1)  # next iteration 3, status "continue"
    assert np.array_equal(lin.view(np.int64), x3.view(np.int64))
