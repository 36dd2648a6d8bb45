#!/usr/bin/env python
"""Time of ONE late iteration of the persistent kernel, by phase: kernel time with 7 and with 10 fixed iterations
(alternating, medians of 15), the slope is a late iteration; LINS_DEBUG_SKIP counting aids drop phases:
  0x10000 solve/update  0x20000 rows  0x40000 row reduction  3 searches  0x80000 certificate tests
  0x100000 de-skew  0x200000 polar view  0x400000 query load
usage: tools/late_iter_time.py [search] [skip ...]"""
import importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
search = sys.argv[1] if len(sys.argv) > 1 else "mr"
skips = [int(a, 0) for a in sys.argv[2:]] or [0, 0x10000, 0x30000, 0x70000, 0x70003, 0xF0003, 0x1F0003]
n = 1024
maxq = int(os.environ.get("AB_MAXQ", "0"))  # > 0: only scans with at most that many queries
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(n * (4 if maxq else 1))))
if maxq:
    pairs = [p for p in pairs if len(p.surf_flat) + len(p.corner_sharp) <= maxq][:n]
os.environ["LINS_ENABLE_DEBUG_KNOBS"] = "1"
os.environ.setdefault("LINS_RELAY_AT", "0")  # whole updates: these are slopes of one scan's chain, not of the launch
print("clock note: times are kernel ms (events); slope = (t10 - t7) / 3")
for skip in skips:
    os.environ["LINS_DEBUG_SKIP"] = str(skip)
    ctxs = {}
    for it in (7, 10):
        c = ieskf.IeskfContext(pkg.default_params(num_iter=it, fixed_iters=1), max_batch=n, max_targets=16384, search=search)
        c.upload(pairs)
        for _ in range(2):
            c.run(); c.sync()
        ctxs[it] = c
    ts = {7: [], 10: []}
    for rep in range(15):
        for it in (7, 10):
            ctxs[it].run(); ctxs[it].sync(); ts[it].append(ctxs[it].last_kernel_ms())
    for c in ctxs.values():
        c.close()
    t7, t10 = np.median(ts[7]), np.median(ts[10])
    d = np.array(ts[10]) - np.array(ts[7])
    print(f"skip {skip:#9x}: t7 {t7:.4f} t10 {t10:.4f} ms  late iteration {(t10 - t7) / 3 * 1e3:6.1f} us (spread of the pairs {np.percentile(d, 25) / 3 * 1e3:.1f}..{np.percentile(d, 75) / 3 * 1e3:.1f})")
