#!/usr/bin/env python
"""Step time of the staged API in three modes (one GPU call): run + sync per step; steps enqueued back to back on one
stream, one sync; the pipelined mode (Joseph kernel beside the next update kernel).  usage: tools/step_modes.py [batch]"""
import importlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = 20
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(n)))
prm = pkg.default_params(num_iter=10, fixed_iters=1)
with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384, search="auto") as c:
    c.upload(pairs)
    for _ in range(3):
        c.run(); c.sync()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            c.run(); c.sync()
        a = (time.perf_counter() - t0) / K * 1e3
        ka = np.mean(c.kernel_ms_history(K))
        t0 = time.perf_counter()
        for _ in range(K):
            c.run()
        c.sync()
        b = (time.perf_counter() - t0) / K * 1e3
        kb = np.mean(c.kernel_ms_history(K))
        c.set_pipelined(True)
        t0 = time.perf_counter()
        for _ in range(K):
            c.run()
        c.sync()
        p = (time.perf_counter() - t0) / K * 1e3
        kp = np.mean(c.kernel_ms_history(K))
        c.set_pipelined(False)
        print(f"sync per step: {a:.4f} ms (kernel {ka:.4f}) | back to back, one stream: {b:.4f} ms (kernel {kb:.4f}) | pipelined: {p:.4f} ms (kernel {kp:.4f})")
