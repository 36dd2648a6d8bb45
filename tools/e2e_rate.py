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
os.environ["LINS_ENABLE_DEBUG_KNOBS"] = "1"
with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384, search="auto") as c:
    arr = defs.pairs_to_c(pairs); res = (defs.ResultC * n)()
    L = ieskf.lib()

    def timed(fn, reps=7):
        for _ in range(2):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    def staged():
        assert L.lins_batch_upload(c._h, n, arr) == 0
        assert L.lins_batch_run(c._h, None, 0) == 0
        assert L.lins_batch_download(c._h, n, res) == 0

    t_up = timed(lambda: L.lins_batch_upload(c._h, n, arr))
    t_st = timed(staged)
    its = n * 10
    print(f"staged upload / run / download, one after the other: upload {t_up*1e3:.2f} ms, whole {t_st*1e3:.2f} ms => {its/t_st/1e6:.2f} M it/s")
    for chunk in (os.environ.get("E2E_CHUNKS", "256").split(",")):
        os.environ["LINS_BATCH_CHUNK"] = chunk
        dt = timed(lambda: L.lins_ieskf_update_batch(c._h, n, arr, res))
        assert sum(r.iters for r in res) == its
        print(f"lins_ieskf_update_batch({n} scans, chunks of {chunk}): {dt*1e3:.2f} ms end-to-end (validate + pack + H2D + kernels + D2H) => {its/dt/1e6:.2f} M it/s")
