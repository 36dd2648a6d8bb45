#!/bin/bash
# round 4, GPU call 12: the feature front-end without the sector sorts (arg-max / arg-min per pick): parity + rate
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_frontend_oracle.py tests/test_gpu_edge_cases.py tests/test_gpu_ref.py tests/test_gpu_sequence.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest12.log 2>&1; tail -5 gpurun_out/r04/pytest12.log
timeout 300 python tools/frontend_rate.py 256 2>&1 | tail -1 | tee gpurun_out/r04/fe_rate12.txt
