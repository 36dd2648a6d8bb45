#!/usr/bin/env python
"""What predicts a workgroup's duration in the batch kernel?  Per-scan phase profile (whole updates) against what is known
before the launch (prior translation) and what is known after the first iteration(s).  usage: tools/duration_predictors.py [batch]"""
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
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
prm = pkg.default_params(num_iter=10, fixed_iters=1)
ctx = ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search="mr")
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
m = prof.astype(float)
total = m[:, 5]
it = m[:, 10:16]  # iterations 0..5 (0..2 only in the default build)
p2 = np.array([float((p.state[:3] ** 2).sum()) for p in pairs])
nq = np.array([len(p.surf_flat) + len(p.corner_sharp) for p in pairs], float)
nt = np.array([len(p.surf_last) + len(p.corner_last) for p in pairs], float)


def r2(x, y):
    c = np.corrcoef(x, y)[0, 1]
    return c * c


print(f"total: mean {total.mean():.0f} ticks, cv {total.std() / total.mean():.3f}")
for name, x in (("|p|^2 (prior translation)", p2), ("queries", nq), ("target points", nt), ("setup", m[:, 0]), ("iteration 0", it[:, 0]),
                ("iterations 0+1", it[:, 0] + it[:, 1]), ("iterations 0..2", it[:, :3].sum(1))):
    rest = total - (m[:, 0] + x if name.startswith("iteration") else 0)
    print(f"  R^2(total, {name}) = {r2(x, total):.3f}" + (f"   R^2(rest of the update, {name}) = {r2(x, rest):.3f}; share of total {x.mean() / total.mean():.2f}"
                                                         if name.startswith("iteration") else ""))
