#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
LINS_IESKF_LIB=$PWD/ab/sgprof.so timeout 300 python - <<'PY' 2>&1 | tail -12 | tee gpurun_out/r04/sg_prof.txt
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
raws = [host.synth_raw_scan(i, 1) for i in range(256)]
with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
    for _ in range(2):
        c.segment_batch(raws)
    print("segment_ms", c.segment_ms())
PY
