#!/usr/bin/env python
"""Shader cycles per call of the serial tail's building blocks (debug_kernels.hip ops 100..111, 64 dependent
repetitions in one 256-thread workgroup per block): round 1's routes next to the round-2 ones.
usage: tools/tail_cycles.py [blocks]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG)
ieskf = importlib.import_module(PKG + ".ieskf")
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
OPS = [(103, "6x6 solve, one lane (reg_solve6)"), (105, "6x6 solve over the wave, elimination + back-substitution (wave_solve6)"),
       (106, "6x6 solve over the wave, Gauss-Jordan (wave_gj_solve6)"), (107, "phi, Rinvleft(-phi)^T: atan2 route (phi_and_Gt_general)"),
       (104, "phi, Rinvleft(-phi)^T: as the kernels call it (series for small rotations)"),
       (108, "axis2quat, libm sin / cos"), (109, "axis2quat_fast"), (110, "quat2axis, libm atan2"), (111, "quat2axis_fast"),
       (100, "transformToStart (one query)"), (101, "plane row"), (102, "28 sums of a wave")]
with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as ctx:
    L = ieskf.lib()
    L.lins_debug_math.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.lins_debug_math.restype = C.c_int
    x = np.array([0.01, 0.02, 0.03, 0.1, 0.2, 0.3, 5.0, 6.0, 1.0])
    for op, name in OPS:
        out = np.zeros((blocks, 2))
        vals = []
        for _ in range(5):
            rc = L.lins_debug_math(ctx._h, op, blocks, x.ctypes.data, 9, out.ctypes.data, 2)
            assert rc == 0, (op, rc)
            vals.append(np.median(out[:, 0]))
        print(f"op {op:3d}  {np.median(vals):9.0f} cycles   {name}")
