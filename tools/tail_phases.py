#!/usr/bin/env python
"""Sub-phases of the serial tail of an iteration (solve_and_update, ieskf_lds_tail.h) on thread 0, shader-clock ticks per
iteration, from a library built with -DLINS_PROF_WAVES=99 -DLINS_PROF_TAIL=1 (tools/build_variant.sh tailprof ...):
system build, Gauss-Jordan solve, dx + boxPlus + staging, barrier, the next iteration's constants, barrier.
usage: LINS_IESKF_LIB=ab/tailprof.so tools/tail_phases.py [batch = 1024] [search = mr]"""
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LINS_ENABLE_DEBUG_KNOBS"] = "1"
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
search = sys.argv[2] if len(sys.argv) > 2 else "mr"
iters = 10
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
ctx = ieskf.IeskfContext(pkg.default_params(num_iter=iters, fixed_iters=1), max_batch=batch, max_targets=16384, search=search)
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
ctx.upload(pairs)
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
for _ in range(2):
    ctx.run(); ctx.sync()
prof = np.zeros((batch, 16), dtype=np.int64)
L.lins_debug_phase_profile(ctx._h, 1, prof.ctypes.data, batch)
names = ["system build", "Gauss-Jordan", "dx + boxPlus + staging", "barrier 1", "next constants", "barrier 2"]
m = prof[:, 6:12].mean(0) / iters
print(f"{search}, {batch} scans x {iters} iterations, kernel {ctx.last_kernel_ms():.4f} ms; tail on thread 0, ticks per iteration (mean over the scans):")
print("  " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, m)) + f"; sum {m.sum():.0f}")
print(f"  the profile's own 'solve' + 'update' slots: {prof[:, 3].mean() / iters:.0f} + {prof[:, 4].mean() / iters:.0f}")
