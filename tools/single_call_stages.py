#!/usr/bin/env python
"""Host-side stage times of lins_ieskf_update for ONE scan pair (what a live filter waits for): the whole call, and
upload / run + wait / download on their own (medians over the 8 stock scans x 5).  usage: tools/single_call_stages.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
defs = importlib.import_module(PKG + "._ctypes_defs")
pairs = [host.synth_pair(i) for i in range(8)]
prm = pkg.default_params(num_iter=30, fixed_iters=0)
with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384, search="auto") as c:
    L = ieskf.lib()
    arrs = [defs.pairs_to_c([p]) for p in pairs]
    res = (defs.ResultC * 1)()
    for a in arrs: L.lins_ieskf_update(c._h, a, res)
    for name, fn in (("update", lambda a: L.lins_ieskf_update(c._h, a, res)),
                     ("upload", lambda a: L.lins_batch_upload(c._h, 1, a)),
                     ("upload+run+sync", lambda a: (L.lins_batch_upload(c._h, 1, a), L.lins_batch_run(c._h, None, 0), L.lins_sync(c._h))),
                     ("run+sync", lambda a: (L.lins_batch_run(c._h, None, 0), L.lins_sync(c._h))),
                     ("download", lambda a: L.lins_batch_download(c._h, 1, res))):
        ts = []
        for rep in range(5):
            for a in arrs:
                t0 = time.perf_counter(); fn(a); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{name:18s} median {ts[len(ts)//2]*1e6:7.1f} us  min {ts[0]*1e6:7.1f} us")
    print("kernel ms", c.last_kernel_ms())
