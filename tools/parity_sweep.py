#!/usr/bin/env python
"""Parity sweep: n seeded scans through every kernel shape (reference stop rule, NUM_ITER 30) vs the CPU oracle
(reduced 6x6 form + kd-tree; the oracle's own tests pin reduced == dense).  Prints mismatch counts."""
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
prm = pkg.default_params(num_iter=30)
with ThreadPoolExecutor(32) as ex:
    pairs = list(ex.map(host.synth_pair, range(start, start + n)))
    t0 = time.time()
    want = list(ex.map(lambda p: oracle.ieskf(prm, p, oracle.FORM_REDUCED, oracle.NN_KDTREE), pairs))
print(f"oracle: {n} scans in {time.time() - t0:.1f} s, iterations {sum(w.iters for w in want)}, diverged {sum(1 for w in want if w.diverged)}")
for search in ("auto", "mr", "split", "lds", "lds1", "binned"):
    with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384, search=search) as c:
        got = c.update_batch(pairs)
    bad_flags = sum((g.iters, g.converged, g.diverged, g.m_surf, g.m_corner) != (w.iters, w.converged, w.diverged, w.m_surf, w.m_corner)
                    for g, w in zip(got, want))
    dp = max(np.abs(g.state[:3] - w.state[:3]).max() for g, w in zip(got, want))
    dq = max(np.abs(g.state[6:10] - w.state[6:10]).max() for g, w in zip(got, want))
    dc = max(np.abs(g.cov - w.cov).max() / np.abs(w.cov).max() for g, w in zip(got, want))
    print(f"{search:7s}: scans with different (iters, converged, diverged, m_surf, m_corner): {bad_flags} of {n}; "
          f"max |dp| {dp:.2e} m, max |dq| {dq:.2e}, max rel |dP| {dc:.2e}")
