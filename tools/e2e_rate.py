#!/usr/bin/env python
"""PCIe-inclusive rate of lins_ieskf_update_batch (host buffers in and out) for DESIGN.md §7."""
import importlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
defs = importlib.import_module(PKG + "._ctypes_defs")
n = 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(n)))
prm = pkg.default_params(num_iter=10, fixed_iters=1)
with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384, search="lds") as c:
    arr = defs.pairs_to_c(pairs); res = (defs.ResultC * n)()
    L = ieskf.lib()
    for _ in range(2):
        assert L.lins_ieskf_update_batch(c._h, n, arr, res) == 0
    t0 = time.perf_counter()
    for _ in range(5):
        assert L.lins_ieskf_update_batch(c._h, n, arr, res) == 0
    dt = (time.perf_counter() - t0) / 5
    its = sum(r.iters for r in res)
    print(f"lins_ieskf_update_batch(1024 scans): {dt*1e3:.2f} ms end-to-end (validate + pack + H2D + kernels + D2H) => {its/dt/1e6:.2f} M it/s")
