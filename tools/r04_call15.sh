#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_frontend_oracle.py tests/test_gpu_sequence.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest15.log 2>&1; tail -3 gpurun_out/r04/pytest15.log
timeout 300 python tools/reproject_rate.py 2>&1 | tail -3 | tee gpurun_out/r04/reproject15.txt
