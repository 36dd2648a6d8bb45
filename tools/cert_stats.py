#!/usr/bin/env python
"""How often the search certificates fire (debug aid). usage: cert_stats.py [n_scans]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG)
host = importlib.import_module(PKG + ".host")
ieskf = importlib.import_module(PKG + ".ieskf")
defs = importlib.import_module(PKG + "._ctypes_defs")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = host.synth_batch(n)
prm = pkg.default_params(num_iter=10, fixed_iters=1)
with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384, search=(sys.argv[2] if len(sys.argv) > 2 else "mr")) as c:
    arr = defs.pairs_to_c(batch)
    res = (defs.ResultC * n)()
    assert ieskf.lib().lins_ieskf_update_batch(c._h, n, arr, res) == 0
    q = np.array([sum(p.sizes()[:2]) for p in batch])
    nn = np.array([r.reserved[1] for r in res]); wk = np.array([r.reserved[2] for r in res])
    print(f"margins cold/warm = {os.environ.get('LINS_MARGIN_COLD','0.10')}/{os.environ.get('LINS_MARGIN_WARM','0.04')}: "
          f"NN searches skipped {nn.sum() / (9 * q.sum()):.1%} of iterations>=1, walks skipped {wk.sum() / (9 * q.sum()):.1%}")
