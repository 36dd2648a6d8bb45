#!/usr/bin/env python
"""Stage times of the device-resident pipeline (lins_streams_step): front-end -> IESKF update -> re-projection."""
import importlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(n)))
    raw0 = list(ex.map(lambda i: host.synth_raw_scan(i, 0), range(n)))
    raw1 = list(ex.map(lambda i: host.synth_raw_scan(i, 1), range(n)))
    seg0 = list(ex.map(host.frontend_segment, raw0))
    seg1 = list(ex.map(host.frontend_segment, raw1))
boot = np.zeros((n, 19))
for i, p in enumerate(pairs):
    boot[i, 0:3], boot[i, 6:10] = p.meta["true_t"], p.meta["true_q"]
st = np.stack([p.state for p in pairs]); cv = np.stack([p.cov for p in pairs])
import ctypes as C
defs = importlib.import_module(PKG + "._ctypes_defs")
cvf = cv.reshape(n, 324)
dp = C.POINTER(C.c_double)
with ieskf.IeskfContext(pkg.default_params(num_iter=10, fixed_iters=1), max_batch=n, max_targets=16384) as c:
    L = ieskf.lib()
    L.lins_streams_step.argtypes = [C.c_void_p, C.POINTER(host.SegmentedScanC), dp, dp, C.c_double, C.POINTER(defs.ResultC), C.POINTER(C.c_int32)]
    L.lins_streams_step_raw.argtypes = [C.c_void_p, C.POINTER(C.POINTER(host.Point)), C.POINTER(C.c_int32), dp, dp, C.c_double, C.POINTER(defs.ResultC), C.POINTER(C.c_int32)]
    res = (defs.ResultC * n)(); counts = np.zeros((n, 4), np.int32); cp = counts.ctypes.data_as(C.POINTER(C.c_int32))
    # (the C arrays are built once: the wall times below are the C call's, not Python's marshalling)
    a0 = (host.SegmentedScanC * n)(*[s.c for s in seg0]); a1 = (host.SegmentedScanC * n)(*[s.c for s in seg1])
    c.streams_init(n)
    c.streams_step(seg0, boot, np.tile(np.eye(18)[None] * 1e-4, (n, 1, 1)))
    best = None
    for rep in range(7):
        t0 = time.perf_counter()
        assert L.lins_streams_step(c._h, a1 if rep % 2 == 0 else a0, st.ctypes.data_as(dp), cvf.ctypes.data_as(dp), 0.1, res, cp) == 0
        wall = time.perf_counter() - t0
        fe, up, rp = c.streams_stats()
        # (the first call allocates the pinned staging; only the steps that feed scan 1 after scan 0 count: the priors are
        # those of that direction, and a step fed the other way round searches more)
        if rep and rep % 2 == 0 and (best is None or wall < best[3]):
            best = (fe, up, rp, wall)
    fe, up, rp, wall = best
    its = sum(r.iters for r in res)
    mb = sum(s.c.n for s in seg0) * 25 / 1e6
    print(f"streams: {n} streams, one scan each: front-end {fe:.3f} ms + update {up:.3f} ms ({its} iterations) + re-projection {rp:.3f} ms "
          f"= {fe + up + rp:.3f} ms on device => {n / (fe + up + rp) * 1e3:.0f} scans/s; the C call incl. validation, packing and the upload of {mb:.0f} MB of segmented clouds: {wall * 1e3:.1f} ms => {n / wall:.0f} scans/s")
    # the same from raw clouds: image projection / segmentation on the device as well
    raws0 = [np.ascontiguousarray(r, dtype=np.float32).reshape(-1, 4) for r in raw0]; raws1 = [np.ascontiguousarray(r, dtype=np.float32).reshape(-1, 4) for r in raw1]
    p0 = (C.POINTER(host.Point) * n)(*[r.ctypes.data_as(C.POINTER(host.Point)) for r in raws0]); p1 = (C.POINTER(host.Point) * n)(*[r.ctypes.data_as(C.POINTER(host.Point)) for r in raws1])
    cn0 = (C.c_int32 * n)(*[len(r) for r in raws0]); cn1 = (C.c_int32 * n)(*[len(r) for r in raws1])
    c.streams_init(n)
    c.streams_step_raw(raw0, boot, np.tile(np.eye(18)[None] * 1e-4, (n, 1, 1)))
    best = None
    for rep in range(7):
        t0 = time.perf_counter()
        assert L.lins_streams_step_raw(c._h, p1 if rep % 2 == 0 else p0, cn1 if rep % 2 == 0 else cn0, st.ctypes.data_as(dp), cvf.ctypes.data_as(dp), 0.1, res, cp) == 0
        wall = time.perf_counter() - t0
        fe, up, rp = c.streams_stats(); sg = c.segment_ms()
        if rep and (best is None or wall < best[4]):
            best = (sg, fe, up, rp, wall)
    sg, fe, up, rp, wall = best
    mb = sum(len(r) for r in raws0) * 16 / 1e6
    print(f"streams from raw clouds: projection + segmentation {sg:.3f} ms + front-end {fe:.3f} ms + update {up:.3f} ms + re-projection {rp:.3f} ms "
          f"= {sg + fe + up + rp:.3f} ms on device => {n / (sg + fe + up + rp) * 1e3:.0f} scans/s; the C call incl. the upload of {mb:.0f} MB of raw clouds: {wall * 1e3:.1f} ms => {n / wall:.0f} scans/s")
