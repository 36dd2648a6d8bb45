#!/usr/bin/env python
"""Kernel time of ONE scan's update (the single-scan kernel, search auto) against the fixed iteration count: set-up, the
cold iteration, the late ones; LINS_DEBUG_SKIP aids as in tools/late_iter_time.py.  usage: tools/single_iter_curve.py [search] [max_iter]"""
import importlib, os, sys
os.environ.setdefault("LINS_ENABLE_DEBUG_KNOBS", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
search = sys.argv[1] if len(sys.argv) > 1 else "auto"
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pairs = [host.synth_pair(i) for i in range(8)]
prev = 0.0
for it in range(1, kmax + 1):
    ts = []
    with ieskf.IeskfContext(pkg.default_params(num_iter=it, fixed_iters=1), max_batch=1, max_targets=16384, search=search) as c:
        for p in pairs:
            c.upload([p])
            for _ in range(2):
                c.run(); c.sync()
            t = []
            for _ in range(7):
                c.run(); c.sync(); t.append(c.last_kernel_ms())
            ts.append(np.median(t))
    m = float(np.mean(ts))
    print(f"{it:2d} iterations: {m * 1e3:7.1f} us   (+{(m - prev) * 1e3:6.1f} us)   [{c.last_search() if False else search}]")
    prev = m
