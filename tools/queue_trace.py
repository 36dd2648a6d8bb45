#!/usr/bin/env python
"""Timeline of a several-part launch of the batch kernel from the per-workgroup trace of a -DLINS_QUEUE_TRACE=1 build
(tools/build_variant.sh qtrace -DLINS_QUEUE_TRACE=1; LINS_IESKF_LIB=ab/qtrace.so): when every workgroup started, how
long it waited for its item (the hand-over of the part before), how long it ran, how many items were running over the
launch, and which updates end it.
usage: tools/queue_trace.py [batch] [num_iter] [fixed 0|1]     (LINS_RELAY_AT / LINS_RELAY_CUTS as debug knobs)"""
import ctypes as C, importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LINS_ENABLE_DEBUG_KNOBS"] = "1"
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fixed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(batch)))
ctx = ieskf.IeskfContext(pkg.default_params(num_iter=iters, fixed_iters=fixed), max_batch=batch, max_targets=16384, search="mr")
L = ieskf.lib()
L.lins_debug_queue_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.upload(pairs)
for _ in range(3):
    ctx.run(); ctx.sync()
res = ctx.download()
parts, _ = ctx.last_cut()
nwg = parts * batch
tr = np.zeros((nwg, 4), dtype=np.int64)
assert L.lins_debug_queue_trace(ctx._h, tr.ctypes.data, nwg) == 0
t0 = tr[:, 0].min()
start, got, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0
item = tr[:, 3]
has = item >= 0
print(f"{nwg} workgroups, kernel (events) {ctx.last_kernel_ms() * 1e3:.0f} us, last end {end.max():.0f} us; knobs at={os.environ.get('LINS_RELAY_AT')} cuts={os.environ.get('LINS_RELAY_CUTS')}")
print(f"  {int(has.sum())} with an item; wait for the item mean {(got - start).mean():.1f} max {(got - start).max():.1f} us; run mean {(end - got)[has].mean():.1f} max {(end - got)[has].max():.1f} us")
edges = np.linspace(0, end.max(), 21)
mid = (edges[:-1] + edges[1:]) / 2
print("  t [us]   :", " ".join(f"{t:5.0f}" for t in mid))
print("  running  :", " ".join(f"{int(((got <= t) & (end > t) & has).sum()):5d}" for t in mid))
print("  resident :", " ".join(f"{int(((start <= t) & (end > t)).sum()):5d}" for t in mid))
part = np.where(has, item >> 27, -1); scan = np.where(has, item & 0x7FFFFFF, -1)
for p in range(parts):
    m = part == p
    if m.any():
        print(f"  part {p}: {m.sum():5d} items, run mean {(end - got)[m].mean():6.1f} max {(end - got)[m].max():6.1f} us, item in hand {got[m].min():.0f}..{got[m].max():.0f} us, ends ..{end[m].max():.0f} us")
it = np.array([r.iters for r in res])
last = np.argsort(end)[-5:]
print("  the five last ends:", " | ".join(f"scan {scan[i]} part {part[i]} {it[scan[i]]} it: in hand {got[i]:.0f}, {end[i] - got[i]:.0f} us" for i in last))
# which scans are the slow ones (round 6): part-0 run time against what the host knows about a scan
p0 = part == 0
dur0 = np.zeros(batch); dur0[scan[p0]] = (end - got)[p0]
rs = lambda c: np.bincount(np.clip(c[:, 3].astype(int), 0, 15), minlength=16)
sizes = np.array([[len(p.surf_flat), len(p.corner_sharp), len(p.surf_last), len(p.corner_last),
                   int(rs(p.surf_last)[:8].sum()), float(np.linalg.norm(p.state[:3]))] for p in pairs], dtype=float)
names = ["n_flat", "n_sharp", "n_less_flat", "n_less_sharp", "less-flat rings 0-7", "|p| prior"]
order = np.argsort(dur0)[::-1]
print("  part-0 run time: mean %.0f, p90 %.0f, p99 %.0f, max %.0f us" % (dur0.mean(), np.percentile(dur0, 90), np.percentile(dur0, 99), dur0.max()))
print("  correlation of the part-0 run time with", ", ".join(f"{n} {np.corrcoef(dur0, sizes[:, k])[0, 1]:+.2f}" for k, n in enumerate(names)))
print("  the ten slowest:", " | ".join(f"scan {s}: {dur0[s]:.0f} us " + "/".join(f"{sizes[s, k]:.0f}" if k < 5 else f"{sizes[s, k]:.2f}" for k in range(6)) for s in order[:10]))
print("  batch means   :", "/".join(f"{sizes[:, k].mean():.0f}" if k < 5 else f"{sizes[:, k].mean():.2f}" for k in range(6)), " resident positions (corner + less-flat rings 0-7): mean %.0f, p90 %.0f, max %.0f" % ((sizes[:, 3] + sizes[:, 4]).mean(), np.percentile(sizes[:, 3] + sizes[:, 4], 90), (sizes[:, 3] + sizes[:, 4]).max()))
