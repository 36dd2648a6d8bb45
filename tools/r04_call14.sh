#!/bin/bash
# round 4, GPU call 14: re-projection fused with the index build (+ short-series rotation, + the relay in the streams step): parity, chain rates
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_frontend_oracle.py tests/test_gpu_edge_cases.py tests/test_gpu_ref.py tests/test_gpu_sequence.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest14.log 2>&1; tail -5 gpurun_out/r04/pytest14.log
timeout 300 python tools/streams_rate.py 1024 2>&1 | tail -4 | tee gpurun_out/r04/streams14.txt
timeout 300 python tools/reproject_rate.py 2>&1 | tail -3 | tee -a gpurun_out/r04/streams14.txt
