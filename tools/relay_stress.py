#!/usr/bin/env python
"""The several-part updates of the batch kernel under repetition and with a COMPETING context (round 6: the ticket protocol
with the release / acquire hand-over): per configuration `launches` lins_batch_run() calls back to back on one context —
a host wait and a download every 100 — while a second context on another stream keeps launching its own batch, so that
the parts of an update meet a changing set of co-resident workgroups; checks that no launch waited out its bound
(lins_last_cut: queue timeouts 0) and that every download returns the first launch's bits.
usage: tools/relay_stress.py [launches = 1200]"""
import importlib, os, sys, threading, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(1500)))


def bits(res):
    return np.concatenate([np.concatenate([np.asarray(r.state), np.asarray(r.cov).ravel(), [r.iters, r.converged, r.diverged, r.m_surf, r.m_corner]]) for r in res])


stop = threading.Event()


def competitor():
    with ieskf.IeskfContext(pkg.default_params(num_iter=10, fixed_iters=1), max_batch=640, max_targets=16384, search="mr") as c:
        c.upload(pairs[700:1340])
        while not stop.is_set():
            for _ in range(20):
                c.run()
            c.sync()


th = threading.Thread(target=competitor, daemon=True)
th.start()
total_timeouts = 0
print(f"# the batch kernel's two launch forms under repetition, a second context launching 640-scan batches all the while; {launches} lins_batch_run per line, a host wait + download every 100")
# (both launch forms: lins_set_launch_queues 1 = every run ONE launch with several-part updates — the hand-over protocol; 2, the
# default = the runs behind the first of each hundred go out as whole-update launches on the context's two launch queues)
for queues, label, prm in [(q, l, p) for q in (1, 2) for l, p in (("fixed 10", pkg.default_params(num_iter=10, fixed_iters=1)), ("stop rule", pkg.default_params(num_iter=30, fixed_iters=0)))]:
    label = f"queues {queues} {label}"
    for n in (1024, 777, 1500):
        with ieskf.IeskfContext(prm, max_batch=n, max_targets=16384, search="mr") as c:
            c.set_launch_queues(queues)
            c.upload(pairs[:n])
            c.run(); c.sync()
            first = bits(c.download())
            same, t0 = True, time.perf_counter()
            for k in range(launches):
                c.run()
                if k % 100 == 99:
                    c.sync()
                    same = same and np.array_equal(bits(c.download()), first, equal_nan=True)
            c.sync()
            dt = time.perf_counter() - t0
            parts, timeouts = c.last_cut()
            total_timeouts += timeouts
            print(f"{label} {n} scans ({parts} parts): {launches} launches in {dt:.2f} s, queue timeouts {timeouts}, same bits as the first launch: {same}", flush=True)
stop.set(); th.join()
print("total queue timeouts:", total_timeouts)
