"""The committed golden fixtures (tests/golden/make_golden.py): the oracle must keep reproducing
them (CPU), the synthetic generator must keep producing their inputs (CPU), and the HIP path must
match them through the C ABI (GPU)."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "pair_*.npz")))


def load(pkg, path):
    z = np.load(path)
    pair = pkg.ScanPair(z["surf_flat"], z["corner_sharp"], z["surf_last"], z["corner_last"], z["state"], z["cov"])
    return z, pair


def test_fixtures_exist():
    assert len(FILES) >= 2


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_generator_reproduces_fixture_inputs(pkg, host, path):
    z, pair = load(pkg, path)
    idx = int(os.path.basename(path).split("_")[1].split(".")[0])
    fresh = host.synth_pair(idx)
    for name in ("surf_flat", "corner_sharp", "surf_last", "corner_last"):
        assert np.array_equal(getattr(fresh, name), getattr(pair, name)), name
    assert np.allclose(fresh.state, pair.state, rtol=0, atol=1e-12) and np.allclose(fresh.cov, pair.cov, rtol=1e-12)


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_oracle_reproduces_golden_outputs(pkg, oracle, path):
    z, pair = load(pkg, path)
    prm = pkg.default_params(num_iter=30)
    for form, nn in ((oracle.FORM_DENSE, oracle.NN_BRUTE), (oracle.FORM_DENSE, oracle.NN_KDTREE),
                     (oracle.FORM_REDUCED, oracle.NN_BRUTE)):
        res, tr = oracle.ieskf(prm, pair, form, nn, trace=True)
        assert [res.iters, res.converged, res.diverged, res.m_surf, res.m_corner] == list(z["out_flags"])
        k = res.iters
        assert np.array_equal(np.stack([tr["surf"][:k][f] for f in ("ind1", "ind2", "ind3")], -1), z["surf_ind"])
        assert np.array_equal(np.stack([tr["corner"][:k][f] for f in ("ind1", "ind2")], -1), z["corner_ind"])
        assert np.array_equal(tr["surf"][:k]["accepted"], z["surf_acc"])
        assert np.array_equal(tr["corner"][:k]["accepted"], z["corner_acc"])
        tol = 0 if form == oracle.FORM_DENSE else 1e-9
        assert np.abs(res.state - z["out_state"]).max() <= tol
        assert np.abs(res.cov - z["out_cov"]).max() <= tol * np.abs(z["out_cov"]).max()
        if form == oracle.FORM_DENSE:
            assert np.array_equal(tr["surf"][:k]["coeff"], z["surf_coeff"])
            assert np.array_equal(tr["corner"][:k]["coeff"], z["corner_coeff"])
    # the estimate lands near the simulated motion
    assert np.linalg.norm(z["out_state"][:3] - z["true_t"]) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("search", ["lds", "lds1", "mr", "auto", "binned", "brute"])
@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_hip_path_matches_golden(pkg, ieskf, path, search):
    z, pair = load(pkg, path)
    prm = pkg.default_params(num_iter=30)
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384, search=search) as ctx:
        got = ctx.update(pair)
        assert [got.iters, got.converged, got.diverged, got.m_surf, got.m_corner] == list(z["out_flags"])
        assert np.abs(got.state[:3] - z["out_state"][:3]).max() <= 1e-6
        assert np.abs(got.state[6:10] - z["out_state"][6:10]).max() <= 1e-7
        assert np.abs(got.cov - z["out_cov"]).max() <= 1e-9 * np.abs(z["out_cov"]).max()
        for k in range(len(z["lin_state"])):
            surf, corner = ctx.correspondences(pair, z["lin_state"][k], k)
            assert np.array_equal(np.stack([surf[f] for f in ("ind1", "ind2", "ind3")], -1), z["surf_ind"][k])
            assert np.array_equal(np.stack([corner[f] for f in ("ind1", "ind2")], -1), z["corner_ind"][k])
            assert np.array_equal(surf["accepted"], z["surf_acc"][k])
            assert np.array_equal(corner["accepted"], z["corner_acc"][k])
            for got_c, want_c in ((surf["coeff"], z["surf_coeff"][k]), (corner["coeff"], z["corner_coeff"][k])):
                ulp = np.abs(got_c.view(np.int32).astype(np.int64) - want_c.view(np.int32).astype(np.int64))
                assert ulp.max(initial=0) <= 1 and (ulp > 0).mean() <= 1e-3
            sums, ms, mc = ctx.reduce_pass(pair, z["lin_state"][k], k)
            assert np.abs(sums - z["sums28"][k]).max() <= 1e-11 * max(1.0, np.abs(z["sums28"][k]).max())
