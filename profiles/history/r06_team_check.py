#!/usr/bin/env python
"""(Belongs to r06_team_of_workgroups.patch — an experiment of round 6 that is NOT in the build.)
One scan on a TEAM of workgroups (one-scan context, search auto) against the one-workgroup kernels: same iteration counts,
flags and row counts, states within rounding of each other (the sums are added up in another order), and the kernel times.
usage (with the patch applied and the library rebuilt): python profiles/history/r06_team_check.py [scans = 64]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pairs = [host.synth_pair(i) for i in range(n)] + [host.synth_pair(70000 + i, scene=1) for i in range(n // 4)]
for label, prm in (("stop rule", pkg.default_params(num_iter=30, fixed_iters=0)), ("fixed 10", pkg.default_params(num_iter=10, fixed_iters=1))):
    out = {}
    for mode in ("auto", "lds", "mr"):
        res, ms = [], []
        with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384, search=mode) as c:
            for p in pairs:
                c.upload([p]); c.run(); c.sync()
                res.append(c.download()[0]); ms.append(c.last_kernel_ms())
            last = c.last_search()
        out[mode] = (res, float(np.mean(ms)), last)
    a = out["auto"][0]
    for other in ("lds", "mr"):
        b = out[other][0]
        flags = sum((x.iters, x.converged, x.diverged, x.m_surf, x.m_corner) != (y.iters, y.converged, y.diverged, y.m_surf, y.m_corner) for x, y in zip(a, b))
        dp = max(float(np.abs(x.state - y.state).max()) for x, y in zip(a, b))
        dP = max(float(np.abs(x.cov - y.cov).max() / np.abs(y.cov).max()) for x, y in zip(a, b))
        print(f"{label}: auto ({out['auto'][2]}) vs {other}: scans with different (iters, flags, row counts) {flags} of {len(a)}; max |d state| {dp:.2e}, max rel |dP| {dP:.2e}")
    print(f"{label}: kernel us per update: auto {out['auto'][1] * 1e3:.1f}, lds {out['lds'][1] * 1e3:.1f}, mr {out['mr'][1] * 1e3:.1f}; mean iterations {np.mean([r.iters for r in a]):.2f}")
