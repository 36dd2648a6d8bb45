#!/usr/bin/env python
"""Kernel-time breakdown of the split path against the persistent kernel at the bench size (GPU box).
Each configuration runs in a fresh process (the debug knobs are read from the environment at lins_create)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, os, sys, time
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, %r)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch, iters, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
prm = pkg.default_params(num_iter=iters, fixed_iters=1)
with ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search=mode) as ctx:
    ctx.upload(pairs)
    for _ in range(3):
        ctx.run(); ctx.sync()
    ks, g, l = [], [], []
    for _ in range(10):
        ctx.run(); ctx.sync(); ks.append(ctx.last_kernel_ms())
        if mode == "split" and iters > 3:
            try:
                a, b = ctx.last_split_ms(); g.append(a); l.append(b)
            except Exception:
                pass
    res = ctx.download()
    print("RESULT", np.mean(ks), np.mean(g) if g else 0, np.mean(l) if l else 0, sum(r.reserved[2] for r in res) / batch)
''' % ROOT


def run(batch, iters, mode, env=None):
    e = dict(os.environ, LINS_ENABLE_DEBUG_KNOBS="1")
    e.update(env or {})
    p = subprocess.run([sys.executable, "-c", CHILD, str(batch), str(iters), mode], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for line in p.stdout.decode().splitlines():
        if line.startswith("RESULT"):
            return [float(x) for x in line.split()[1:]]
    return [float("nan")] * 4 + [p.stderr.decode()[-300:]]


batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for it in (1, 2, 3, 4, 6, 10):
    r = run(batch, it, "mr")
    print(f"mr    iters {it:2d}: kernel {r[0]:.3f} ms")
for name, env in [("default", {}), ("no exhaustive", {"LINS_DEBUG_SKIP": "64"}), ("no lists+no exh.", {"LINS_DEBUG_SKIP": "192"}),
                  ("split_iters 2", {"LINS_SPLIT_ITERS": "2"}), ("split_iters 4", {"LINS_SPLIT_ITERS": "4"}),
                  ("margin 0.05", {"LINS_SPLIT_MARGIN": "0.05"}), ("margin 0.2", {"LINS_SPLIT_MARGIN": "0.2"}),
                  ("margin 0.4", {"LINS_SPLIT_MARGIN": "0.4"})]:
    r = run(batch, 10, "split", env)
    print(f"split {name:18s}: kernel {r[0]:.3f} ms = grid {r[1]:.3f} + list {r[2]:.3f}; exhaustive searches per scan {r[3]:.2f}", r[4:] if len(r) > 4 else "")
