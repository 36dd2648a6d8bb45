#!/usr/bin/env python
"""Kernel time / HBM rate of the updatePointCloud re-projection (lins_transform_to_end_batch) for DESIGN.md §7."""
import importlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
n = 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(n)))
clouds = [p.surf_last for p in pairs] + [p.corner_last for p in pairs]
rng = np.random.default_rng(0)
poses = [(rng.normal(0, 0.3, 3), np.array([1.0, 0.01, -0.02, 0.015]) / np.linalg.norm([1.0, 0.01, -0.02, 0.015])) for _ in clouds]
with ieskf.IeskfContext(pkg.default_params(), max_batch=n, max_targets=16384) as c:
    for yzx in (True, False):
        best = (1e9, 0)
        for _ in range(4):
            t0 = time.perf_counter()
            c.transform_to_end(clouds, poses, yzx=yzx)
            wall = time.perf_counter() - t0
            ms, b = c.reproject_stats()
            best = min(best, (ms, b))
        ms, b = best
        npts = sum(len(cl) for cl in clouds)
        print(f"transform_to_end_batch yzx={yzx}: {len(clouds)} clouds, {npts} points, kernel {ms:.3f} ms, "
              f"{b/ms/1e6:.1f} GB/s of 8000 ({b/ms/1e6/8000:.3f}), {npts/ms/1e6:.2f} G points/s; wall incl. PCIe+pack {wall*1e3:.1f} ms")
