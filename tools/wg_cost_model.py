#!/usr/bin/env python
"""How well do a scan's sizes predict its workgroup's run time?  (input to the longest-first launch order)
usage: tools/wg_cost_model.py [batch] [search]"""
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
    pairs = list(ex.map(host.synth_pair, range(batch)))
prm = pkg.default_params(num_iter=10, fixed_iters=1)
ctx = ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search=search)
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
ctx.upload(pairs)
for _ in range(2):
    ctx.run(); ctx.sync()
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
ctx.run(); ctx.sync()
prof = np.zeros((batch, 16), dtype=np.int64)
L.lins_debug_phase_profile(ctx._h, 1, prof.ctypes.data, batch)
total = prof[:, 5].astype(float)
sizes = np.array([p.sizes() for p in pairs], dtype=float)  # n_surf_q, n_corner_q, n_surf_t, n_corner_t
X = np.concatenate([sizes, np.ones((batch, 1))], 1)
coef, *_ = np.linalg.lstsq(X, total, rcond=None)
pred = X @ coef
r2 = 1 - ((total - pred) ** 2).sum() / ((total - total.mean()) ** 2).sum()
print("ticks: mean %.0f min %.0f max %.0f std %.0f" % (total.mean(), total.min(), total.max(), total.std()))
print("linear model on (n_surf_q, n_corner_q, n_surf_t, n_corner_t, 1):", np.round(coef, 1), "R^2 = %.3f" % r2)
for k, nm in enumerate(["n_surf_q", "n_corner_q", "n_surf_t", "n_corner_t"]):
    print(f"  corr(total, {nm}) = {np.corrcoef(total, sizes[:, k])[0, 1]:.3f}   mean {sizes[:, k].mean():.0f} min {sizes[:, k].min():.0f} max {sizes[:, k].max():.0f}")
ctx.run(); ctx.sync()
prof2 = np.zeros((batch, 16), dtype=np.int64)
L.lins_debug_phase_profile(ctx._h, 1, prof2.ctypes.data, batch)
print("corr(run 1, run 2) of per-scan ticks = %.3f" % np.corrcoef(total, prof2[:, 5].astype(float))[0, 1])
st = np.array([p.state for p in pairs])
feats = {"|v|": np.linalg.norm(st[:, 3:6], axis=1), "|p|": np.linalg.norm(st[:, 0:3], axis=1)}
for nm, f in feats.items():
    print(f"  corr(total, {nm}) = {np.corrcoef(total, f)[0, 1]:.3f}")
for k, nm in enumerate(["setup", "corr", "reduce", "solve", "update"]):
    print(f"  corr(total, {nm} ticks) = {np.corrcoef(total, prof[:, k].astype(float))[0, 1]:.3f}")
start = prof[:, 14].astype(float); end = prof[:, 15].astype(float)
t0 = start.min()
print("wall clock (100 MHz): first start 0, last start %.1f us, last end %.1f us; sum of durations / 512 slots = %.1f us" %
      ((start.max() - t0) / 100, (end.max() - t0) / 100, (end - start).sum() / 512 / 100))

import heapq
def simulate(order, dur, slots=512):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for i in order:
        t = heapq.heappop(h) + dur[i]
        end = max(end, t)
        heapq.heappush(h, t)
    return end
dur = (end - start) / 100.0
v = feats["|v|"]
print("list-scheduling model, 512 slots: as submitted %.0f us; by |v| descending %.0f us; by measured duration descending %.0f us; ideal %.0f us" %
      (simulate(range(batch), dur), simulate(np.argsort(-v), dur), simulate(np.argsort(-dur), dur), dur.sum() / 512))
# what the host can know before the launch: the four cloud sizes
cd, *_ = np.linalg.lstsq(X, dur, rcond=None)
print("  by the linear size model descending %.0f us (model coefficients per point [us]: %s); by n_surf_t descending %.0f us; by queries descending %.0f us" %
      (simulate(np.argsort(-(X @ cd)), dur), np.round(cd, 4), simulate(np.argsort(-sizes[:, 2]), dur), simulate(np.argsort(-(sizes[:, 0] + sizes[:, 1])), dur)))
print("  duration: mean %.1f us, std %.1f, min %.1f, max %.1f" % (dur.mean(), dur.std(), dur.min(), dur.max()))

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
res = ctx.download() if hasattr(ctx, "download") else None
np.savez(os.path.join(ROOT, "gpurun_out", "wg_durations.npz"), dur=dur, prof=prof)
