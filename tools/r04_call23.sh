#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
for v in fe_prev fe_merged fe_prev fe_merged; do echo -n "$v: "; LINS_IESKF_LIB=$PWD/ab/$v.so timeout 300 python tools/frontend_rate.py 256 2>&1 | tail -1 | cut -c1-110; done | tee gpurun_out/r04/fe_ab23.txt
timeout 600 python -m pytest tests/test_frontend_oracle.py tests/test_gpu_edge_cases.py tests/test_gpu_ref.py tests/test_gpu_sequence.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest23.log 2>&1; tail -2 gpurun_out/r04/pytest23.log
