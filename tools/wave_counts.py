#!/usr/bin/env python
"""Window scans of the batch kernel's searches, per wave (library built with -DLINS_PROF2=k -DLINS_MR_CAP=3920: the iterations >= k):
how many scan_spans() calls and how many grid positions the busiest lane of a wave goes through in the nearest-neighbour phase and
in the walk phase, and the sums over the lanes — is a phase's time its calls (dependent LDS round trips) or its points?
usage: LINS_IESKF_LIB=ab/prof0c.so tools/wave_counts.py [iterations = 1]"""
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
ctx = ieskf.IeskfContext(pkg.default_params(num_iter=iters, fixed_iters=1), max_batch=batch, max_targets=16384, search="mr")
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
L.lins_debug_wave_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L.lins_debug_wave_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.upload(pairs)
ctx.run(); ctx.sync()
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
ctx.run(); ctx.sync()
ph = np.zeros((batch, 8, 8), np.int32); cn = np.zeros((batch, 8, 8), np.int32)
assert L.lins_debug_wave_phases(ctx._h, ph.ctypes.data, batch) == 0 and L.lins_debug_wave_counts(ctx._h, cn.ctypes.data, batch) == 0
print(f"iterations 0..{iters - 1}, mean over {batch} workgroups; waves 0-4 plane queries, 5-7 line queries")
print("wave |  nn ticks   calls(max lane)  points(max lane)  calls(sum)  points(sum) | walk ticks   calls(max)  points(max)  calls(sum)  points(sum)")
for w in range(8):
    c = cn[:, w].mean(0)
    print(f"{w:4d} | {ph[:, w, 1].mean():9.0f} {c[0]:12.1f} {c[1]:16.1f} {c[2]:12.0f} {c[3]:12.0f} | {ph[:, w, 2].mean():9.0f} {c[4]:12.1f} {c[5]:12.1f} {c[6]:11.0f} {c[7]:11.0f}")
for name, t, c, p in (("nn", ph[:, :, 1], cn[:, :, 0], cn[:, :, 1]), ("walk", ph[:, :, 2], cn[:, :, 4], cn[:, :, 5])):
    t = t.reshape(-1).astype(float); A = np.stack([c.reshape(-1), p.reshape(-1), np.ones(t.size)], 1).astype(float)
    k, *_ = np.linalg.lstsq(A, t, rcond=None)
    r2 = 1 - ((A @ k - t) ** 2).sum() / ((t - t.mean()) ** 2).sum()
    print(f"{name}: ticks ~ {k[0]:.0f} x calls(max lane) + {k[1]:.1f} x points(max lane) + {k[2]:.0f}   (R^2 {r2:.2f} over wave x workgroup)")
