#!/usr/bin/env python
"""Per-wave correspondence-phase time of the batch kernel (PROF variant built with -DLINS_PROF_WAVES=k: iterations >= k):
how far the slowest wave of a workgroup is from the average one.  usage: LINS_IESKF_LIB=ab/pw0.so tools/wave_balance.py"""
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch = 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
prm = pkg.default_params(num_iter=10, fixed_iters=1)
ctx = ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search="mr")
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
ctx.upload(pairs)
for _ in range(2):
    ctx.run(); ctx.sync()
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
ctx.run(); ctx.sync()
prof = np.zeros((batch, 16), dtype=np.int64)
L.lins_debug_phase_profile(ctx._h, 1, prof.ctypes.data, batch)
w = prof[:, 6:14].astype(float)
tot = prof[:, 5].astype(float)
print("per-wave correspondence ticks, mean over workgroups:", np.round(w.mean(0) / 1e3, 1), "k")
print("mean over waves %.1f k, max over waves %.1f k (mean over workgroups) => the slowest wave is %.2f x the average; "
      "which wave is slowest: %s" % (w.mean(1).mean() / 1e3, w.max(1).mean() / 1e3, (w.max(1) / w.mean(1)).mean(),
                                      np.bincount(w.argmax(1), minlength=8)))
print("workgroup total %.1f k ticks; (max - mean) over waves = %.1f %% of it" % (tot.mean() / 1e3, 100 * (w.max(1) - w.mean(1)).mean() / tot.mean()))
slow = np.argsort(-tot)[:10]
print("ten slowest workgroups: total", np.round(tot[slow] / 1e3), "max wave", np.round(w[slow].max(1) / 1e3), "mean wave", np.round(w[slow].mean(1) / 1e3))
