#!/usr/bin/env python
"""A/B kernel (and whole-step wall) timing of two builds of liblins_ieskf.so in ONE GPU call (boxes differ by several per cent, so two
calls cannot be compared): alternates the libraries, several rounds, prints the medians.
usage: tools/ab_timing.py libA.so libB.so[:ENV=VAL[,ENV=VAL]] [mode ...]   (modes default: mr)
A library may carry debug-knob settings after a colon (LINS_ENABLE_DEBUG_KNOBS=1 is set for it), e.g.
  tools/ab_timing.py lib.so lib.so:LINS_LAUNCH_ORDER=0 lib.so:LINS_MARGIN_COLD=0.1,LINS_MARGIN_WARM=0.04"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, os, sys
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, %r)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch, mode = int(sys.argv[1]), sys.argv[2]
maxq = int(os.environ.get("AB_MAXQ", "0"))  # > 0: only scans with at most that many queries (experiments with smaller workgroups)
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch * (4 if maxq else 1))))
if maxq:
    pairs = [p for p in pairs if len(p.surf_flat) + len(p.corner_sharp) <= maxq][:batch]
    assert len(pairs) == batch, len(pairs)
prm = pkg.default_params(num_iter=10, fixed_iters=1)
with ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search=mode) as ctx:
    ctx.upload(pairs)
    for _ in range(3):
        ctx.run(); ctx.sync()
    import time
    ks, ws = [], []
    for _ in range(15):
        t0 = time.perf_counter(); ctx.run(); ctx.sync(); ws.append((time.perf_counter() - t0) * 1e3); ks.append(ctx.last_kernel_ms())
    print("WALL %%.4f" %% float(np.median(ws)))
    print("RESULT %%.4f" %% float(np.median(ks)))
''' % ROOT


WALL = {}


def run(lib, mode, batch=1024):
    path, _, knobs = lib.partition(":")
    e = dict(os.environ, LINS_IESKF_LIB=os.path.abspath(path))
    if knobs:
        e["LINS_ENABLE_DEBUG_KNOBS"] = "1"
        for kv in knobs.split(","):
            k, _, v = kv.partition("=")
            e[k] = v
    p = subprocess.run([sys.executable, "-c", CHILD, str(batch), mode], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = float("nan")
    for line in p.stdout.decode().splitlines():
        if line.startswith("WALL"):
            wall = float(line.split()[1])
        if line.startswith("RESULT"):
            WALL.setdefault((lib, mode), []).append(wall)
            return [float(x) for x in line.split()[1:]]
    return [float("nan"), p.stderr.decode()[-200:]]


libs = [a for a in sys.argv[1:] if '.so' in a]
modes = [a for a in sys.argv[1:] if '.so' not in a] or ["mr"]
res = {(l, m): [] for l in libs for m in modes}
for rnd in range(3):
    for m in modes:
        for l in libs:
            res[(l, m)].append(run(l, m))
for m in modes:
    for l in libs:
        r = np.array([x[0] if isinstance(x[0], float) else np.nan for x in res[(l, m)]], dtype=float)
        print(f"{m:6s} {os.path.basename(l):60s} kernel {np.nanmedian(r):.4f} ms   step (run + sync, wall) {np.nanmedian(WALL.get((l, m), [np.nan])):.4f} ms")
