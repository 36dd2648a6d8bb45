#!/usr/bin/env python
"""Kernel time / rate of the scan-to-map row (lins_scan2map_batch) next to the CPU oracle (oracle_scan2map, 1 core)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); ieskf = importlib.import_module(PKG + ".ieskf"); defs = importlib.import_module(PKG + "._ctypes_defs")
from map_synth import make_problem
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
probs = [make_problem(defs, 100 + i, n_map_surf=30000, n_map_corner=4000, n_scan_surf=1500, n_scan_corner=400)[0] for i in range(n)]
t0 = time.perf_counter()
want = [oracle.scan2map(p) for p in probs[:4]]
cpu = (time.perf_counter() - t0) / 4
with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        got = c.scan2map_batch(probs)
        wall = time.perf_counter() - t0
        ms, q = c.map_stats()
        if best is None or ms < best[0]:
            best = (ms, q, wall)
    ms, q, wall = best
    # the same local maps again (the mapping node's map only changes with its key frames): resident, not re-uploaded
    for p in probs:
        p.reuse_resident_map = True
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        got2 = c.scan2map_batch(probs)
        walls.append(time.perf_counter() - t0)
    ms2, _ = c.map_stats()
    assert all(a["iters"] == b["iters"] and (a["transform"] == b["transform"]).all() for a, b in zip(got, got2))
    print(f"scan-to-map, maps resident (LINS_MAP_REUSE): device sequence {ms2:.3f} ms, whole call {min(walls) * 1e3:.2f} ms "
          f"(= {min(walls) * 1e3 / ms2:.2f} x the device time)")
    same = all(g["iters"] == w["iters"] and g["n_sel"] == w["n_sel"] for g, w in zip(got, want))
    rounds = sum(g["iters"] for g in got)
    print(f"scan-to-map: {n} problems (map 30000 surf + 4000 corner, scan 1500 + 400), {rounds} rounds, {q} query evaluations: "
          f"device sequence (10 rounds: correspondence + step kernels) {ms:.3f} ms = {q / ms / 1e3:.1f} M queries/s; whole call {wall * 1e3:.1f} ms = {wall / n * 1e3:.2f} ms/problem; "
          f"oracle {cpu * 1e3:.1f} ms/problem on 1 core => {cpu / (wall / n):.1f}x; rounds/selected rows equal the oracle's on the first 4: {same}")
