#!/usr/bin/env python
"""Parity sweep: n seeded scans through every kernel shape (reference stop rule, NUM_ITER 30) vs the CPU oracle
(reduced 6x6 form + kd-tree; the oracle's own tests pin reduced == dense).  Prints mismatch counts.
usage: tools/parity_sweep.py [n] [first] [wide] [open]"""
import importlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
start = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
# third argument: a prior that MATTERS.  The shipped filter starts from init_pos_std = init_att_std = 0, so P_SS is tiny
# and the posterior state hardly depends on the 28 sums; "wide" adds a seeded, fully correlated SPD block (5 cm / 0.5 deg
# / 0.1 m/s and matching bias scales) to every prior covariance, so that gain, solve and Joseph update carry weight.
wide = "wide" in sys.argv[3:]
scene = 1 if "open" in sys.argv[3:] else 0  # ("open": the second scene family of csrc/host/synth.cpp)
prm = pkg.default_params(num_iter=30)
with ThreadPoolExecutor(32) as ex:
    pairs = list(ex.map(lambda i: host.synth_pair(i, scene=scene), range(start, start + n)))
if scene:
    sz = np.array([p.sizes() for p in pairs])
    print(f"# scene family 'open': mean sizes (sharp, flat, lessSharp, lessFlat) {np.round(sz.mean(0), 1)}")
if wide:
    scale = np.array([0.05] * 3 + [0.1] * 3 + [0.009] * 3 + [0.02] * 3 + [0.002] * 3 + [0.01] * 3)
    for k, p in enumerate(pairs):
        m = np.random.default_rng(900000 + start + k).normal(size=(18, 18)) / np.sqrt(18.0)
        p.cov = np.ascontiguousarray(p.cov + (scale[:, None] * (m @ m.T + 0.5 * np.eye(18)) * scale[None, :]))
    print("# prior covariances widened (argument 'wide'): P += D (M M^T + I/2) D, D = diag(5 cm, 0.1 m/s, 0.5 deg, ...)")
with ThreadPoolExecutor(32) as ex:
    t0 = time.time()
    want = list(ex.map(lambda p: oracle.ieskf(prm, p, oracle.FORM_REDUCED, oracle.NN_KDTREE), pairs))
print(f"oracle: {n} scans in {time.time() - t0:.1f} s, iterations {sum(w.iters for w in want)}, diverged {sum(1 for w in want if w.diverged)}")
for search in ("auto", "mr", "lds", "lds1", "binned"):
    with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384, search=search) as c:
        got = c.update_batch(pairs)
    bad_flags = sum((g.iters, g.converged, g.diverged, g.m_surf, g.m_corner) != (w.iters, w.converged, w.diverged, w.m_surf, w.m_corner)
                    for g, w in zip(got, want))
    dp = max(np.abs(g.state[:3] - w.state[:3]).max() for g, w in zip(got, want))
    dq = max(np.abs(g.state[6:10] - w.state[6:10]).max() for g, w in zip(got, want))
    dc = max(np.abs(g.cov - w.cov).max() / np.abs(w.cov).max() for g, w in zip(got, want))
    print(f"{search:7s}: scans with different (iters, converged, diverged, m_surf, m_corner): {bad_flags} of {n}; "
          f"max |dp| {dp:.2e} m, max |dq| {dq:.2e}, max rel |dP| {dc:.2e}")
