#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
LINS_IESKF_LIB=$PWD/ab/prof0n.so timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/r04/nn_subphases.txt
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.getcwd())
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch = 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
ctx = ieskf.IeskfContext(pkg.default_params(num_iter=1, fixed_iters=1), max_batch=batch, max_targets=16384, search="mr")
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
L.lins_debug_wave_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L.lins_debug_wave_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.upload(pairs); ctx.run(); ctx.sync()
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
ctx.run(); ctx.sync()
ph = np.zeros((batch, 8, 8), np.int32); cn = np.zeros((batch, 8, 8), np.int32)
assert L.lins_debug_wave_phases(ctx._h, ph.ctypes.data, batch) == 0 and L.lins_debug_wave_counts(ctx._h, cn.ctypes.data, batch) == 0
print("cold iteration, mean over 1024 workgroups (ticks on lane 0 of each wave)")
print("wave | nn phase | coop_nn calls | inside nn_lds: seeds  task tests  window scans  lane merge | nn_lds whole")
for w in range(8):
    c = cn[:, w].mean(0)
    print(f"{w:4d} | {ph[:, w, 1].mean():8.0f} | {c[7]:6.2f} | {c[1]:8.0f} {c[2]:8.0f} {c[3]:8.0f} {c[4]:8.0f} | {c[5]:8.0f}")
PY
