#!/usr/bin/env python
"""How many workgroups (= scans) are resident per CU at a time: from the PROF variant's HW_ID / wall-clock probe.
usage: tools/residency.py [batch] [search]"""
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
search = sys.argv[2] if len(sys.argv) > 2 else "mr3"
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
ctx = ieskf.IeskfContext(pkg.default_params(num_iter=10, fixed_iters=1), max_batch=batch, max_targets=16384, search=search)
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
ctx.upload(pairs)
for _ in range(2):
    ctx.run(); ctx.sync()
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
ctx.run(); ctx.sync()
prof = np.zeros((batch, 16), dtype=np.int64)
L.lins_debug_phase_profile(ctx._h, 1, prof.ctypes.data, batch)
hw = prof[:, 13] & 0xFFFFFFFF; xcc = (prof[:, 13] >> 32) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0, t1 = prof[:, 14], prof[:, 15]
print(f"{search}: kernel {ctx.last_kernel_ms():.3f} ms, distinct CUs {len(np.unique(key))}, WG wall time mean {np.mean(t1 - t0) / 100:.1f} us")
maxc = []
for k in np.unique(key):
    m = key == k
    ev = sorted([(a, 1) for a in t0[m]] + [(b, -1) for b in t1[m]])
    c = mx = 0
    for _, d in ev:
        c += d; mx = max(mx, c)
    maxc.append(mx)
print("max concurrently resident WGs per CU: histogram", np.bincount(maxc))
