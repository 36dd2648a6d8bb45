#!/usr/bin/env python
"""Per-wave, per-phase shader-clock ticks of the batch kernel's late iterations (library built with -DLINS_PROF2=k: the
iterations >= k; whole updates — the phase profile switches the cuts off): where a late iteration's time goes on every
wave, which wave the barrier waits for and in which phase it fell behind.
usage: LINS_IESKF_LIB=ab/prof2.so tools/wave_phases.py [k = 5] [iterations = 10]"""
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
k0 = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = 1024
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
prm = pkg.default_params(num_iter=iters, fixed_iters=1)
ctx = ieskf.IeskfContext(prm, max_batch=batch, max_targets=16384, search="mr")
L = ieskf.lib()
L.lins_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
L.lins_debug_wave_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.upload(pairs)
for _ in range(2):
    ctx.run(); ctx.sync()
L.lins_debug_phase_profile(ctx._h, 1, None, 0)
ctx.run(); ctx.sync()
ph = np.zeros((batch, 8, 8), dtype=np.int32)
assert L.lins_debug_wave_phases(ctx._h, ph.ctypes.data, batch) == 0
n = iters - k0
# (phase slot 4 carries the search counts instead of the ~230-tick reduction: wave-iterations with a walk | walks << 12 | wave-iterations with a nearest-neighbour search << 20 | searches << 26)
cnt = ph[:, :, 4].copy().view(np.uint32)
ph[:, :, 4] = 0
w_it, w_n, n_it, n_n = cnt & 0xFFF, (cnt >> 12) & 0xFF, (cnt >> 20) & 0x3F, cnt >> 26
p = ph.astype(float) / n  # ticks per late iteration
names = ["load+deskew", "nn", "walk", "rows", "reduce", "barrierA", "fold+barB", "solve/upd"]
print(f"ticks per late iteration (iterations {k0}..{iters - 1}), mean over {batch} workgroups; waves 0-4 plane queries, 5-7 line queries")
print("wave   " + "".join(f"{x:>12s}" for x in names) + "       total")
for w in range(8):
    print(f"{w:4d}   " + "".join(f"{p[:, w, j].mean():12.0f}" for j in range(8)) + f"{p[:, w].sum(1).mean():12.0f}")
work = p[:, :, :5].sum(2)  # a wave's own work before the barrier
print("own work before the barrier (phases 0-4): mean over waves %.0f, max over waves %.0f (mean over workgroups); slowest wave: %s" %
      (work.mean(), work.max(1).mean(), np.bincount(work.argmax(1), minlength=8)))
for s_ in (np.argsort(p[:, 0].sum(1))[::-1][:3] if os.environ.get("WP_SLOWEST") else []):  # (round 6) the slowest workgroups, wave by wave
    print(f"  scan {s_} (|p| prior {np.linalg.norm(pairs[s_].state[:3]):.2f}): ticks per iteration by wave")
    for w in range(8):
        print(f"  {w:4d}   " + "".join(f"{p[s_, w, j]:12.0f}" for j in range(8)) + f"{p[s_, w].sum():12.0f}")
slow = work.argmax(1)
sel = p[np.arange(batch), slow]
print("the slowest wave's phases: " + ", ".join(f"{names[j]} {sel[:, j].mean():.0f}" for j in range(5)))
fast = work.argmin(1)
self_ = p[np.arange(batch), fast]
print("the fastest wave's phases: " + ", ".join(f"{names[j]} {self_[:, j].mean():.0f}" for j in range(5)))
it = p[:, 0].sum(1)
print("one late iteration on wave 0: %.0f ticks = own work %.0f + barrier wait %.0f + fold %.0f + solve/update %.0f" %
      (it.mean(), p[:, 0, :5].sum(1).mean(), p[:, 0, 5].mean(), p[:, 0, 6].mean(), p[:, 0, 7].mean()))
for thr in (1.25, 1.5, 2.0):
    print("  workgroups whose slowest wave needs > %.2f x the mean wave: %.1f %%" % (thr, 100 * (work.max(1) > thr * work.mean(1)).mean()))
print("nn phase of the slowest wave vs of the others: %.0f vs %.0f; walk: %.0f vs %.0f; rows: %.0f vs %.0f" % (
    sel[:, 1].mean(), p[:, :, 1].mean(), sel[:, 2].mean(), p[:, :, 2].mean(), sel[:, 3].mean(), p[:, :, 3].mean()))
print("per wave: share of late wave-iterations that run a nearest-neighbour search | a walk; searches | walks per such wave-iteration")
for w in range(8):
    print(f"  wave {w}: nn {n_it[:, w].mean() / n:.3f} | walk {w_it[:, w].mean() / n:.3f}; {n_n[:, w].sum() / max(1, n_it[:, w].sum()):.2f} | {w_n[:, w].sum() / max(1, w_it[:, w].sum()):.2f}")
has = (w_it > 0)
print("walk phase of wave-iterations: workgroups where wave 4 never walked in the late iterations: %d; its walk phase there %.0f ticks, elsewhere %.0f" % (
    (~has[:, 4]).sum(), p[~has[:, 4], 4, 2].mean() if (~has[:, 4]).any() else float('nan'), p[has[:, 4], 4, 2].mean()))
# per-query view (LINS_PROF2 builds leave (searches, walks, walk mask, ring) of every query in its slot)
nslots = sum(len(q.surf_flat) + len(q.corner_sharp) for q in pairs)
L.lins_debug_query_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
qs = np.zeros((nslots, 4), dtype=np.int32)
assert L.lins_debug_query_slots(ctx._h, qs.ctypes.data, nslots) == 0
is_surf = np.concatenate([np.r_[np.ones(len(q.surf_flat), bool), np.zeros(len(q.corner_sharp), bool)] for q in pairs])
for name, m in (("plane", is_surf), ("line", ~is_surf)):
    w = qs[m, 1]
    print(f"{name} queries: {m.sum()}; walks in the {n} late iterations: mean {w.mean():.3f} per query; queries with 0 / 1 / 2 / 3+ / all {n}: "
          f"{(w == 0).mean():.3f} / {(w == 1).mean():.3f} / {(w == 2).mean():.3f} / {(w >= 3).mean():.3f} / {(w == n).mean():.4f}; "
          f"share of the walks that come from queries walking in EVERY late iteration: {w[w == n].sum() / max(1, w.sum()):.2f}; from 3+: {w[w >= 3].sum() / max(1, w.sum()):.2f}")
    nnc = np.array([bin(int(x) >> 16).count("1") for x in qs[m, 2]])
    print(f"   walks that follow a change of the nearest neighbour: {nnc.sum() / max(1, w.sum()):.2f} of them; nearest-neighbour searches per query {qs[m, 0].mean():.3f}")
    if name == "plane":
        ring = qs[m, 3] & 0xFF
        for r in range(8):
            mm = ring == r
            if mm.sum():
                print(f"   ring {r}: {mm.sum():6d} queries, walks per query and late iteration {w[mm].mean() / n:.3f}")
