#!/usr/bin/env python
"""The slowest updates of the batch (round 6): per-workgroup phase clocks of the PROF variant (whole updates), the
slowest scans against the mean — total, set-up, correspondence phase, the first three iterations — next to what the host
knows about them.  usage: tools/slow_scans.py [batch] [search]"""
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
search = sys.argv[2] if len(sys.argv) > 2 else "mr"
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map((lambda i: host.synth_pair(i, scene=1)) if os.environ.get("SLOW_SCENE_B") else host.synth_pair, range(50000, 50000 + batch) if os.environ.get("SLOW_SCENE_B") else range(batch)))
ctx = ieskf.IeskfContext(pkg.default_params(num_iter=10, fixed_iters=1), max_batch=batch, max_targets=16384, search=search)
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
wall = (prof[:, 15] - prof[:, 14]) / 100.0
names = ["setup", "corr", "reduce", "solve", "update", "total", "t0 deskew", "t0 nn", "t0 walk", "t0 rows", "iter 0", "iter 1", "iter 2"]
pn = np.array([float(np.linalg.norm(p.state[:3])) for p in pairs])
vn = np.array([float(np.linalg.norm(p.state[3:6])) for p in pairs])
print(f"{search}: kernel {ctx.last_kernel_ms():.3f} ms; workgroup wall time mean {wall.mean():.0f} p90 {np.percentile(wall, 90):.0f} p99 {np.percentile(wall, 99):.0f} max {wall.max():.0f} us")
print("ticks (shader clock), mean over the batch:", ", ".join(f"{n} {prof[:, k].mean():.0f}" for k, n in enumerate(names)))
order = np.argsort(wall)[::-1]
for s in order[:8]:
    print(f"  scan {s}: wall {wall[s]:.0f} us |p| {pn[s]:.2f} |v| {vn[s]:.2f} queries {len(pairs[s].surf_flat)}+{len(pairs[s].corner_sharp)} rows {res[s].m_surf}+{res[s].m_corner}: " +
          ", ".join(f"{n} {prof[s, k]}" for k, n in enumerate(names) if k in (0, 1, 3, 5, 10, 11, 12)))
print("correlation of the wall time with |p| %.2f, |v| %.2f, iter 0 %.2f, iter 1 %.2f, iter 2 %.2f" % tuple(np.corrcoef(wall, x)[0, 1] for x in (pn, vn, prof[:, 10], prof[:, 11], prof[:, 12])))
late = prof[:, 5] - prof[:, 0] - prof[:, 10] - prof[:, 11] - prof[:, 12]
print("iterations 3-9 + epilogue, ticks: mean %.0f, slowest eight %s" % (late.mean(), [int(late[s]) for s in order[:8]]))
