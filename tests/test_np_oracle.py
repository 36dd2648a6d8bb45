"""The two CPU restatements of the reference — oracle/lins_oracle.cpp (C++) and oracle/np_oracle.py (numpy, written
independently, no shared code) — must agree on the golden pairs: index triplets and accepted sets bit for bit, the
f32 rows to the last ulp, the posterior to 1e-9.  The only substitute available for reference vectors (the
reference ships none and cannot be built here): a misreading would have to be made twice to survive."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "pair_*.npz")))


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_numpy_restatement_equals_cpp_oracle(pkg, oracle, path):
    from oracle import np_oracle

    z = np.load(path)
    pair = pkg.ScanPair(z["surf_flat"], z["corner_sharp"], z["surf_last"], z["corner_last"], z["state"], z["cov"])
    prm = pkg.default_params(num_iter=30)  # the shipped NUM_ITER, reference stop rule
    want, tr = oracle.ieskf(prm, pair, oracle.FORM_DENSE, oracle.NN_BRUTE, trace=True)
    got = np_oracle.perform_ieskf(prm, pair)
    assert (got["iters"], got["converged"], got["diverged"], got["m_surf"], got["m_corner"]) == \
        (want.iters, want.converged, want.diverged, want.m_surf, want.m_corner)
    n_vals = n_off = 0
    for it in range(want.iters):
        for kind in ("surf", "corner"):
            g, w = got["trace"][it][kind], tr[kind][it]
            assert np.array_equal(g["ind"][:, 0], w["ind1"]), (it, kind)
            # (the oracle reports second / third indices only when the first exists; -1 otherwise on both sides)
            assert np.array_equal(g["ind"][:, 1], w["ind2"]), (it, kind)
            if kind == "surf":
                assert np.array_equal(g["ind"][:, 2], w["ind3"]), (it, kind)
            assert np.array_equal(g["acc"], w["accepted"]), (it, kind)
            for a, b in ((g["coeff"], w["coeff"]), (g["sel"][:, :3], w["sel"][:, :3])):
                d = _ulp_diff(a, b)
                assert d.max() <= 1, (it, kind, d.max())  # same expression types; libm / summation order: last ulp
                n_vals += d.size
                n_off += int((d > 0).sum())
    assert n_off <= 1e-3 * n_vals, (n_off, n_vals)
    assert np.abs(got["state"] - want.state).max() <= 1e-9
    assert np.abs(got["cov"] - want.cov).max() <= 1e-9 * np.abs(want.cov).max()


def test_numpy_restatement_identities():
    """Hand-checkable properties of the second restatement's own helpers (not shared with the C++ oracle)."""
    from oracle import np_oracle as o

    rng = np.random.default_rng(5)
    for _ in range(20):
        v = rng.normal(size=3) * rng.choice([1e-12, 1e-3, 0.3, 0.9])
        v = v * min(1.0, 3.0 / max(np.linalg.norm(v), 1e-300))  # (|v| < pi: Quat2axis wraps beyond)
        q = o.axis2quat(v)
        assert abs(np.linalg.norm(q) - 1) < 1e-15
        if np.linalg.norm(v) >= 1e-10:
            assert np.allclose(o.quat2axis(q), v, atol=1e-12)
            # Rinvleft(phi) is the inverse of the left Jacobian of SO(3)
            th = np.linalg.norm(v)
            a = v / th
            jl = np.sin(th) / th * np.eye(3) + (1 - np.sin(th) / th) * np.outer(a, a) + (1 - np.cos(th)) / th * o.skew(a)
            assert np.allclose(o.rinvleft(v) @ jl, np.eye(3), atol=1e-9)
        assert np.allclose(o.qmatrix(q) @ np.array([1.0, 2.0, 3.0]), o.qrotate(q, np.array([1.0, 2.0, 3.0])), atol=1e-14)
    s = np.zeros(19)
    s[6] = 1.0
    s[18] = -9.81
    dx = rng.normal(size=18) * 0.1
    assert np.allclose(o.box_minus(o.box_plus(s, dx), s), dx, atol=1e-12)


def test_diverging_pair_on_both_cpu_restatements(pkg, oracle):
    """(CPU) the C++ oracle and the independent numpy restatement agree that the pair takes SE:566-570."""
    from diverging import make_diverging_pair
    from oracle import np_oracle

    pair = make_diverging_pair(pkg)
    prm = pkg.default_params(num_iter=30)
    a = oracle.ieskf(prm, pair, oracle.FORM_DENSE, oracle.NN_BRUTE)
    b = np_oracle.perform_ieskf(prm, pair)
    assert (a.iters, a.converged, a.diverged, a.m_surf) == (b["iters"], b["converged"], b["diverged"], b["m_surf"]) == (2, 0, 1, 20)


@pytest.mark.parametrize("seed", [3, 11])
def test_both_restatements_with_a_prior_that_matters(pkg, host, oracle, seed):
    """The shipped filter starts from init_pos_std = init_att_std = 0: P_SS is tiny and the posterior hardly depends on
    the 28 sums, so agreement on the goldens says little about the gain / solve / Joseph algebra.  Here every prior gets
    a seeded, fully correlated 18 x 18 block (5 cm / 0.5 deg / 0.1 m/s, as `tools/parity_sweep.py ... wide` on the GPU):
    the posterior moves by centimetres, and the two independently written restatements must still agree — with each
    other, and the C++ oracle's dense M x M form with its reduced 6 x 6 form."""
    from oracle import np_oracle

    pair = host.synth_pair(700 + seed)
    scale = np.array([0.05] * 3 + [0.1] * 3 + [0.009] * 3 + [0.02] * 3 + [0.002] * 3 + [0.01] * 3)
    m = np.random.default_rng(seed).normal(size=(18, 18)) / np.sqrt(18.0)
    pair.cov = np.ascontiguousarray(pair.cov + scale[:, None] * (m @ m.T + 0.5 * np.eye(18)) * scale[None, :])
    prm = pkg.default_params(num_iter=30)
    dense = oracle.ieskf(prm, pair, oracle.FORM_DENSE, oracle.NN_BRUTE)
    red = oracle.ieskf(prm, pair, oracle.FORM_REDUCED, oracle.NN_KDTREE)
    got = np_oracle.perform_ieskf(prm, pair)
    for r in (red,):
        assert (r.iters, r.converged, r.diverged, r.m_surf, r.m_corner) == (dense.iters, dense.converged, dense.diverged, dense.m_surf, dense.m_corner)
        assert np.abs(r.state - dense.state).max() <= 1e-10 and np.abs(r.cov - dense.cov).max() <= 1e-12 * np.abs(dense.cov).max()
    assert (got["iters"], got["converged"], got["diverged"], got["m_surf"], got["m_corner"]) == \
        (dense.iters, dense.converged, dense.diverged, dense.m_surf, dense.m_corner)
    assert np.abs(got["state"] - dense.state).max() <= 1e-9
    assert np.abs(got["cov"] - dense.cov).max() <= 1e-9 * np.abs(dense.cov).max()
    assert np.abs(dense.state[:3] - pair.state[:3]).max() > 1e-3  # the prior does matter: the position moved by millimetres at least
