#!/usr/bin/env python
"""What the end of a launch costs: the same 1024-scan batch run from TWO contexts (two streams) whose launches overlap — the
slots one launch leaves idle at its end are taken by the other's workgroups — against one context.  Wall time per launch.
(A measurement for DESIGN.md, not a mode of bench.py: per-kernel durations are not defined when launches overlap.)"""
import importlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG)
host = importlib.import_module(PKG + ".host")
ieskf = importlib.import_module(PKG + ".ieskf")

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
prm = pkg.default_params(num_iter=10, fixed_iters=1)
ctxs = [ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search="mr") for _ in range(2)]
for c in ctxs:
    c.upload(pairs)
    c.run()
    c.sync()
for name, use in (("one context", ctxs[:1]), ("two contexts, launches alternated", ctxs)):
    for rep in range(3):
        n = 200
        t0 = time.perf_counter()
        for k in range(n):
            use[k % len(use)].run()
        for c in use:
            c.sync()
        dt = time.perf_counter() - t0
        print(f"{name}: {dt / n * 1e3:.4f} ms per launch ({batch * 10 * n / dt / 1e6:.2f} M it/s)", flush=True)
