#!/usr/bin/env python
"""Set-up and the cold iteration of the batch kernel, taken apart with the LINS_DEBUG_SKIP counting aids:
kernel time of 1 (and 2) fixed iterations with the nearest-neighbour searches (2), the index walks (1), both (3) or
every phase (0x5f0003: what is left is the set-up + an empty loop) dropped.  usage: tools/cold_iter_time.py"""
import importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
n = 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(n)))
os.environ["LINS_ENABLE_DEBUG_KNOBS"] = "1"
os.environ.setdefault("LINS_RELAY_AT", "0")  # whole updates: these are slopes of one scan's chain, not of the launch
cfgs = [(1, 0), (1, 3), (1, 2), (1, 1), (1, 0x5F0003), (2, 0), (2, 3)]
ctxs = []
for it, skip in cfgs:
    os.environ["LINS_DEBUG_SKIP"] = str(skip)
    c = ieskf.IeskfContext(pkg.default_params(num_iter=it, fixed_iters=1), max_batch=n, max_targets=16384, search="mr")
    c.upload(pairs)
    for _ in range(2):
        c.run(); c.sync()
    ctxs.append(c)
ts = np.zeros((11, len(cfgs)))
for rep in range(11):
    for k, c in enumerate(ctxs):
        c.run(); c.sync(); ts[rep, k] = c.last_kernel_ms()
for (it, skip), t in zip(cfgs, np.median(ts, axis=0)):
    print(f"iterations {it} skip {skip:#x}: {t:.4f} ms")
