#!/usr/bin/env python
"""The batch launch under the reference's stop rule (NUM_ITER 30, |dx| <= 1e-2): when do the scans that never converge
start and end, and what does one of their iterations cost?  Whole updates with the phase-profile variant (one record per
workgroup: start / end on the 100 MHz wall clock).
usage: tools/stop_rule_timeline.py [batch]"""
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
prm = pkg.default_params(num_iter=30)
ctx = ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search="mr")
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
ctx.upload(pairs)
for _ in range(2):
    ctx.run(); ctx.sync()
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
ctx.run(); ctx.sync()
prof = np.zeros((batch, 16), dtype=np.int64)
L.lins_debug_phase_profile(ctx._h, 1, prof.ctypes.data, batch)
res = ctx.download()
it = np.array([r.iters for r in res])
start = prof[:, 14].astype(float) / 100; end = prof[:, 15].astype(float) / 100
t0 = start.min(); start -= t0; end -= t0
dur = end - start
print(f"launch: last end {end.max():.0f} us; sum of durations / 512 slots {dur.sum() / 512:.0f} us; kernel (events) {ctx.last_kernel_ms() * 1e3:.0f} us")
print("iterations: histogram", dict(zip(*np.unique(it, return_counts=True))))
for lo, hi in ((1, 5), (6, 9), (10, 29), (30, 30)):
    m = (it >= lo) & (it <= hi)
    if m.any():
        print(f"  scans with {lo}..{hi} iterations: {m.sum():4d}  duration mean {dur[m].mean():6.0f} max {dur[m].max():6.0f} us   start mean {start[m].mean():5.0f} max {start[m].max():5.0f}   end mean {end[m].mean():5.0f} max {end[m].max():5.0f}")
m30 = it == 30
if m30.any():
    short = dur[(it >= 4) & (it <= 8)]
    print(f"a 30-iteration update: {dur[m30].mean():.0f} us mean, {dur[m30].min():.0f}..{dur[m30].max():.0f}; the work of the batch it is: {dur[m30].sum() / dur.sum():.2f}")
    # cost of a late iteration of such a scan: (duration - duration of a typical 6-iteration scan) / 24
    six = dur[it == 6].mean() if (it == 6).any() else short.mean()
    print(f"  per late iteration ~ {(dur[m30].mean() - six) / 24:.1f} us (against {six:.0f} us for a 6-iteration update)")
    late = np.sort(end[m30])[-5:]
    print("  the five last ends:", np.round(late), " their starts:", np.round(start[m30][np.argsort(end[m30])[-5:]]))
