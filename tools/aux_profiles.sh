#!/bin/bash
# rocprofv3 kernel stats of the rows around the update (ICP fallback, re-projection, feature front-end, segmentation, scan-to-map)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cat > /tmp/aux_run.py <<'PY'
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
from concurrent.futures import ThreadPoolExecutor
n = 256
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(n)))
    raws = list(ex.map(lambda i: host.synth_raw_scan(i, 1), range(n)))
    segs = list(ex.map(host.frontend_segment, raws))
rng = np.random.default_rng(0)
bad = []
for p in pairs:
    st = p.state.copy(); st[0:3] += rng.normal(0, 0.15, 3)
    bad.append(pkg.ScanPair(p.surf_flat, p.corner_sharp, p.surf_last, p.corner_last, st, p.cov))
with ieskf.IeskfContext(pkg.default_params(num_iter=30), max_batch=n, max_targets=16384) as c:
    for _ in range(3):
        r = c.icp_update_batch(bad)
        c.extract_features_batch(segs)
        c.segment_batch(raws)
        c.transform_to_end([p.surf_last for p in pairs] + [p.corner_last for p in pairs],
                           [(np.array([0.3, 0.1, 0.0]), np.array([0.9999, 0.01, 0.0, 0.01]) / np.linalg.norm([0.9999, 0.01, 0.0, 0.01]))] * (2 * n))
    sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
    from map_synth import make_problem
    defs = importlib.import_module(PKG + "._ctypes_defs")
    probs = [make_problem(defs, 100 + i, n_map_surf=30000, n_map_corner=4000, n_scan_surf=1500, n_scan_corner=400)[0] for i in range(32)]
    for _ in range(3):
        c.scan2map_batch(probs)
    print("ICP rounds per scan: mean", np.mean([x.iters for x in r]), "converged", sum(x.converged for x in r), "of", n)
PY
cd /tmp && export TMPDIR=/tmp && rm -rf $out/aux_kt && rocprofv3 --kernel-trace --stats -d $out/aux_kt -- python /tmp/aux_run.py > $out/aux_kt.log 2>&1
grep "ICP rounds" $out/aux_kt.log
python $root/tools/rocpd_summary.py $(find $out/aux_kt -name "*.db") > $out/aux_kernel_stats.csv; rm -rf $out/aux_kt
cat $out/aux_kernel_stats.csv
