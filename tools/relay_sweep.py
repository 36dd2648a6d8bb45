#!/usr/bin/env python
"""Kernel time of the batch kernel against the iterations its updates are cut at (LINS_RELAY_AT, LINS_RELAY_CUTS: cuts at
at, 2 at, ... cuts x at; 0 = whole updates).
usage: tools/relay_sweep.py [batch] [at[:cuts] ...]   (one process, contexts alternated: boxes differ by a few per cent)"""
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
ats = sys.argv[2:] or ["0", "3", "4", "5", "6", "7"]
iters = int(os.environ.get("RS_ITERS", "10"))
fixed = int(os.environ.get("RS_FIXED", "1"))
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
prm = pkg.default_params(num_iter=iters, fixed_iters=fixed)
os.environ["LINS_ENABLE_DEBUG_KNOBS"] = "1"
ctxs = {}
for at in ats:
    os.environ["LINS_RELAY_AT"] = at.partition(":")[0]
    f = at.split(":")  # at[:cuts]
    os.environ.pop("LINS_RELAY_CUTS", None)
    if len(f) > 1:
        os.environ["LINS_RELAY_CUTS"] = f[1]
    c = ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search="mr")
    c.upload(pairs)
    ctxs[at] = c
ref = None
times = {at: [] for at in ats}
for rep in range(4):
    for at in ats:
        c = ctxs[at]
        for _ in range(8):
            c.run()
        c.sync()
        times[at] += c.kernel_ms_history(8)[2:]
for at in ats:
    res = ctxs[at].download()
    key = [(r.iters, r.converged, r.diverged, r.m_surf, r.m_corner, r.state.tobytes(), r.cov.tobytes()) for r in res]
    if ref is None:
        ref = key
    same = key == ref
    t = np.array(times[at])
    print(f"relay_at {at}: kernel {t.mean():.4f} ms (min {t.min():.4f}, {len(t)} launches)  same bits as the first: {same}")
