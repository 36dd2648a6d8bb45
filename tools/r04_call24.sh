#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
LINS_IESKF_LIB=$PWD/ab/prof5n.so timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/r04/solve_subphases.txt
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.getcwd())
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch = 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
ctx = ieskf.IeskfContext(pkg.default_params(num_iter=10, fixed_iters=1), max_batch=batch, max_targets=16384, search="mr")
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
L.lins_debug_wave_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L.lins_debug_wave_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.upload(pairs); ctx.run(); ctx.sync()
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
ctx.run(); ctx.sync()
ph = np.zeros((batch, 8, 8), np.int32); cn = np.zeros((batch, 8, 8), np.int32)
assert L.lins_debug_wave_phases(ctx._h, ph.ctypes.data, batch) == 0 and L.lins_debug_wave_counts(ctx._h, cn.ctypes.data, batch) == 0
print("late iterations 5..9, ticks per iteration, mean over 1024 workgroups")
print("solve/update phase of wave 0 (prof2[7]):", ph[:, 0, 7].mean() / 5)
for w in range(3):
    print(f"wave {w}: from the phase's start to the first barrier {cn[:, w, 0].mean() / 5:.0f} (wave 0: solve_wave0), barrier + next_iter_consts {cn[:, w, 6].mean() / 5:.0f}")
PY
