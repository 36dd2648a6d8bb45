#!/usr/bin/env python
"""Measures BASELINE.json configs[0..3] on this box and prints the BASELINE.md result rows."""
import importlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG)
host = importlib.import_module(PKG + ".host")
ieskf = importlib.import_module(PKG + ".ieskf")
from oracle import oracle  # noqa: E402  (cpu_baseline leg)

prm = pkg.default_params(num_iter=10, fixed_iters=1)
pair = host.synth_pair(0)
ncpu = os.cpu_count()

# configs[0]: CPU oracle, single scan, 10 iterations
reps = 20
sec, its = oracle.bench(prm, [pair] * reps, oracle.FORM_DENSE, oracle.NN_KDTREE, threads=1)
cpu1 = its / sec
print(f"| 1. single scan, CPU oracle, 10 iters | {cpu1:.0f} | n/a (one scan) | n/a | n/a | self |")

with ieskf.IeskfContext(prm, max_batch=1, max_targets=16384, search="auto") as ctx:
    # configs[1]: device correspondences + reduction, host 18x18 solve, per iteration
    want, tr = oracle.ieskf(prm, pair, oracle.FORM_DENSE, oracle.NN_BRUTE, trace=True)
    lin = pair.state.copy()
    t0 = time.perf_counter()
    n = 0
    for rep in range(5):
        lin = pair.state.copy()
        for k in range(10):
            sums, ms, mc = ctx.reduce_pass(pair, lin, k)
            dx, _, _ = ieskf.host_solve_from_sums(prm, pair, lin, sums)
            lin = oracle.box_plus(lin, dx)  # (host algebra of the harness; not timed as product)
            n += 1
    t1 = time.perf_counter()
    dp = np.abs(lin[:3] - want.state[:3]).max()
    print(f"| 2. single scan, 1 GPU, host 18x18 solve | {cpu1:.0f} | - | {n / (t1 - t0):.0f} (incl. PCIe upload per call) | - | idx exact, |dp|={dp:.1e} |")
    # configs[2]: full on-device loop
    for _ in range(3):
        r = ctx.update(pair)
    t0 = time.perf_counter()
    for _ in range(50):
        r = ctx.update(pair)
    t1 = time.perf_counter()
    ctx.upload([pair])
    ks = []
    for _ in range(20):
        ctx.run()
        ctx.sync()
        ks.append(ctx.last_kernel_ms())
    kms = float(np.median(ks))
    bpi = pair.bytes_per_iter()
    print(f"| 3. single scan, 1 GPU, on-device loop | {cpu1:.0f} | - | {50 * r.iters / (t1 - t0):.0f} end-to-end (H2D+D2H), "
          f"{r.iters / (kms * 1e-3):.0f} kernel-only ({kms * 1e3:.0f} us / {r.iters} iters) | "
          f"{100 * bpi * r.iters / (kms * 1e-3) / 8e12:.4f} % | state 1e-6/1e-7, cov 1e-9 |")

with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(1024)))
with ieskf.IeskfContext(prm, max_batch=1024, max_targets=16384, search="auto") as ctx:
    ctx.upload(pairs)
    ks = []
    for _ in range(10):
        ctx.run()
        ctx.sync()
        ks.append(ctx.last_kernel_ms())
    kms = float(np.median(ks[2:]))
    it = ctx.total_iters()
    sample = pairs[:128]
    secn, itn = oracle.bench(prm, sample, oracle.FORM_DENSE, oracle.NN_KDTREE, threads=ncpu)
    print(f"| 4. batch 1024, 1 GPU | {cpu1:.0f} | {itn / secn:.0f} ({ncpu} threads) | {it / (kms * 1e-3):.0f} | "
          f"{100 * ctx.bytes_per_iter() / 1024 * it / (kms * 1e-3) / 8e12:.2f} % | flags/m exact, state 1e-6/1e-7 |")
