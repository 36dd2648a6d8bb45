#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per run, kernel-trace only) of the kernels around the update:
# segmentation, feature front-end, re-projection, scan-to-map.  Writes gpurun_out/aux_pmc.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cat > /tmp/aux_pmc_run.py <<'PY'
import importlib, os, sys
import numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
defs = importlib.import_module(PKG + "._ctypes_defs")
from concurrent.futures import ThreadPoolExecutor
from map_synth import make_problem
n = 256
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(n)))
    raws = list(ex.map(lambda i: host.synth_raw_scan(i, 1), range(n)))
    segs = list(ex.map(host.frontend_segment, raws))
probs = [make_problem(defs, 100 + i, n_map_surf=30000, n_map_corner=4000, n_scan_surf=1500, n_scan_corner=400)[0] for i in range(32)]
with ieskf.IeskfContext(pkg.default_params(), max_batch=n, max_targets=16384) as c:
    for _ in range(2):
        c.segment_batch(raws)
        c.extract_features_batch(segs)
        c.transform_to_end([p.surf_last for p in pairs], [(np.array([0.3, 0.1, 0.0]), np.array([1.0, 0.0, 0.0, 0.0]))] * n)
        c.scan2map_batch(probs)
print("points: raw", sum(len(r) for r in raws), "segmented", sum(s.n for s in segs), "re-projected", sum(len(p.surf_last) for p in pairs))
PY
cd /tmp && export TMPDIR=/tmp
: > $out/aux_pmc.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/aux_pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d $out/aux_pmc_$ctr -- python /tmp/aux_pmc_run.py > $out/aux_pmc_$ctr.log 2>&1
  grep "^points" $out/aux_pmc_$ctr.log >> $out/aux_pmc.txt
  python $root/tools/rocpd_summary.py $(find $out/aux_pmc_$ctr -name "*.db") | grep -E "$ctr|^kernel,counter" | sed 's/void lins:://; s/lins:://; s/([^)]*)//' >> $out/aux_pmc.txt
  rm -rf $out/aux_pmc_$ctr
done
cat $out/aux_pmc.txt
