#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q --maxfail=10 -k "streams or one_kernel" > gpurun_out/r04/pytest16.log 2>&1; tail -5 gpurun_out/r04/pytest16.log
