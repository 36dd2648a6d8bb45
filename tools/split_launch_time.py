#!/usr/bin/env python
"""Step time of the staged API (20 lins_batch_run queued back to back, one wait; best of 5) with one launch per run
(several-part updates) against two launch queues (lins_set_launch_queues), on both scene families, fixed iterations and
the reference's stop rule.  usage: tools/split_launch_time.py [batch]"""
import importlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with ThreadPoolExecutor(16) as ex:
    room = list(ex.map(host.synth_pair, range(n)))
    open_ = list(ex.map(lambda i: host.synth_pair(i, scene=1), range(50000, 50000 + n)))
for label, pairs in (("room", room), ("open scene", open_)):
    for mode, prm in (("fixed 10", pkg.default_params(num_iter=10, fixed_iters=1)), ("stop rule", pkg.default_params(num_iter=30, fixed_iters=0))):
        row = []
        for queues in (1, 2):
            with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384, search="mr") as c:
                c.set_launch_queues(queues)
                c.upload(pairs)
                for _ in range(3):
                    c.run()
                c.sync()
                best, one = 1e9, 1e9
                for rep in range(5):
                    t0 = time.perf_counter()
                    for _ in range(20):
                        c.run()
                    c.sync()
                    best = min(best, (time.perf_counter() - t0) / 20)
                    t0 = time.perf_counter(); c.run(); c.sync(); one = min(one, time.perf_counter() - t0)
                row.append((best * 1e3, one * 1e3))
        print(f"{label:10s} {mode:9s} {n} scans: one launch {row[0][0]:.4f} ms/step queued, {row[0][1]:.4f} ms one step + wait | two queues {row[1][0]:.4f} ms/step queued, {row[1][1]:.4f} ms one step + wait")
