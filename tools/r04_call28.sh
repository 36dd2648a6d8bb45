#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
for v in sg_prev sg_new sg_prev sg_new; do echo -n "$v: "; LINS_IESKF_LIB=$PWD/ab/$v.so timeout 300 python - <<'PY' 2>&1 | tail -1
import importlib, os, sys
sys.path.insert(0, os.getcwd())
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
raws = [host.synth_raw_scan(i, 1) for i in range(256)]
with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
    ms = []
    for _ in range(4):
        c.segment_batch(raws); ms.append(c.segment_ms())
    print("segment_ms per 256 scans: %.4f" % min(ms))
PY
done | tee gpurun_out/r04/sg_ab28.txt
timeout 600 python -m pytest tests/test_frontend_oracle.py tests/test_gpu_edge_cases.py tests/test_gpu_ref.py tests/test_gpu_sequence.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest28.log 2>&1; tail -2 gpurun_out/r04/pytest28.log
