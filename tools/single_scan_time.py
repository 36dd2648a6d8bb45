#!/usr/bin/env python
"""Single-scan latency (configs[2]): kernel time of one scan through the full-residency kernel under the reference's
stop rule, medians over several scans.  usage: tools/single_scan_time.py [search]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
search = sys.argv[1] if len(sys.argv) > 1 else "auto"
pairs = [host.synth_pair(i) for i in range(8)]
with ieskf.IeskfContext(pkg.default_params(num_iter=30), max_batch=1, max_targets=16384, search=search) as c:
    ms, its = [], []
    for p in pairs:
        c.upload([p])
        for _ in range(3):
            c.run(); c.sync()
        t = []
        for _ in range(9):
            c.run(); c.sync(); t.append(c.last_kernel_ms())
        ms.append(np.median(t)); its.append(c.download()[0].iters)
    print(f"single scan, search {search}: kernel {np.mean(ms) * 1e3:.1f} us per update (mean of 8 scans, {np.mean(its):.1f} iterations each) = {np.sum(ms) / np.sum(its) * 1e3:.1f} us per iteration")
