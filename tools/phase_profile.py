#!/usr/bin/env python
"""Per-phase shader-clock profile of the persistent IESKF kernel (debug aid).
usage: tools/phase_profile.py [batch] [search]"""
import ctypes as C
import importlib
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG)
host = importlib.import_module(PKG + ".host")
ieskf = importlib.import_module(PKG + ".ieskf")

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
search = sys.argv[2] if len(sys.argv) > 2 else "binned"
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
prm = pkg.default_params(num_iter=int(os.environ.get("PP_ITERS", "10")), fixed_iters=1)
ctx = ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search=search)
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
ctx.upload(pairs)
for _ in range(2):
    ctx.run()
    ctx.sync()
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
ctx.run()
ctx.sync()
prof = np.zeros((batch, 16), dtype=np.int64)
L.lins_debug_phase_profile(ctx._h, 1, prof.ctypes.data, batch)
print(f"batch {batch} search {search}: kernel ms", ctx.last_kernel_ms())
names = ["setup", "corr", "reduce", "solve", "update", "total", "t0/deskew", "t0/nn", "t0/walk", "t0/geom", "iter0", "iter1", "iter2", "iter3", "iter4", "iter5"]
m = prof[:, :16].astype(float)
for i, n in enumerate(names):
    print(f"{n:7s} mean {m[:, i].mean():12.0f}  min {m[:, i].min():12.0f}  max {m[:, i].max():12.0f} ticks")
# effective shader clock: total ticks of a workgroup / its wall time (100 MHz counter), and the launch's span
wall = (prof[:, 15] - prof[:, 14]).astype(float)
ok = wall > 0
print("effective shader clock: %.3f GHz (median over workgroups)" % np.median(m[ok, 5] / (wall[ok] * 10.0)))
span = (prof[:, 15].max() - prof[:, 14].min()) * 10e-9 * 1e3
print("launch span (first start .. last end): %.3f ms; sum of workgroup wall times / %d slots: %.3f ms" % (span, 512, wall.sum() * 10e-9 * 1e3 / 512))
