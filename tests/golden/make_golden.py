#!/usr/bin/env python
"""Regenerates tests/golden/*.npz.

The reference has no golden vectors for this path, so these are THIS repo's goldens:
seeded synthetic scan pairs (inputs) with the CPU oracle's outputs (dense MxM form, brute-force
NN): final state / covariance / flags and the per-iteration correspondence triplets + f32 rows.
They pin the oracle against silent drift and give the GPU parity tests inputs that do not depend
on the generator.  Since round 3 the REFERENCE'S OWN CODE (oracle/_ref: StateEstimator.hpp
compiled verbatim) is run on the same inputs and must agree before a file is written
(tests/test_ref.py::test_reference_reproduces_the_committed_goldens checks the committed files).

Regenerated in round 3: the synthetic sensor's firings now carry a seeded azimuth phase + jitter
(csrc/host/synth.cpp) instead of sitting exactly on image_projection_node's column edges.

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"


def main():
    import __graft_entry__ as g

    g.build()
    pkg = importlib.import_module(PKG)
    host = importlib.import_module(PKG + ".host")
    from oracle import oracle, ref

    out_dir = os.path.dirname(os.path.abspath(__file__))
    prm = pkg.default_params(num_iter=30)
    for idx in (0, 3):
        pair = host.synth_pair(idx)
        res, tr = oracle.ieskf(prm, pair, oracle.FORM_DENSE, oracle.NN_BRUTE, trace=True)
        k = res.iters
        if ref.available():  # the reference's own sources agree before anything is written
            r = ref.perform_ieskf(prm, pair)
            assert (r.iters, r.converged, r.diverged, r.m_surf, r.m_corner) == (res.iters, res.converged, res.diverged, res.m_surf, res.m_corner)
            assert np.abs(r.state - res.state).max() <= 1e-12 and np.abs(r.cov - res.cov).max() <= 1e-12 * np.abs(res.cov).max()
            for it in range(k):
                s, c = ref.correspondences(prm, pair, tr["lin_state"][it], it)
                for got, want in ((s, tr["surf"][it]), (c, tr["corner"][it])):
                    assert all(np.array_equal(got[f], want[f]) for f in ("ind1", "ind2", "ind3", "accepted"))
                    assert np.array_equal(got["coeff"].view(np.int32), want["coeff"].view(np.int32))
        np.savez_compressed(
            os.path.join(out_dir, f"pair_{idx}.npz"),
            surf_flat=pair.surf_flat, corner_sharp=pair.corner_sharp, surf_last=pair.surf_last,
            corner_last=pair.corner_last, state=pair.state, cov=pair.cov,
            true_t=pair.meta["true_t"], true_q=pair.meta["true_q"],
            out_state=res.state, out_cov=res.cov,
            out_flags=np.array([res.iters, res.converged, res.diverged, res.m_surf, res.m_corner]),
            out_residual_norm=res.residual_norm,
            lin_state=tr["lin_state"][:k], dx=tr["dx"][:k], sums28=tr["sums28"][:k],
            surf_ind=np.stack([tr["surf"][:k]["ind1"], tr["surf"][:k]["ind2"], tr["surf"][:k]["ind3"]], -1),
            surf_acc=tr["surf"][:k]["accepted"], surf_coeff=tr["surf"][:k]["coeff"],
            corner_ind=np.stack([tr["corner"][:k]["ind1"], tr["corner"][:k]["ind2"]], -1),
            corner_acc=tr["corner"][:k]["accepted"], corner_coeff=tr["corner"][:k]["coeff"],
        )
        print(f"pair_{idx}.npz: sizes {pair.sizes()} iters {res.iters} m=({res.m_surf},{res.m_corner})")


if __name__ == "__main__":
    main()
