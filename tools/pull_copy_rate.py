#!/usr/bin/env python
"""Host -> device rate of the pinned staging arena: hipMemcpyAsync (what the uploads do) against a copy kernel that reads the
host memory over PCIe (lins_debug_pull_copy).  usage: tools/pull_copy_rate.py [MB ...]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); ieskf = importlib.import_module(PKG + ".ieskf")
sizes = [int(a) for a in sys.argv[1:]] or [4, 36, 147]
with ieskf.IeskfContext(pkg.default_params(), max_batch=1024, max_targets=16384) as c:
    L = ieskf.lib()
    L.lins_debug_pull_copy.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double)]
    for mb in sizes:
        r = []
        for mode in (0, 1):
            g = C.c_double(0)
            assert L.lins_debug_pull_copy(c._h, mb << 20, 5, mode, C.byref(g)) == 0
            r.append(g.value)
        print(f"{mb:4d} MB: hipMemcpyAsync {r[0]:.1f} GB/s, copy kernel over the mapped pointer {r[1]:.1f} GB/s")
