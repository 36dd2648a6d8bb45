#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q --maxfail=10 -k "rough or wide_hall" > gpurun_out/r04/pytest26.log 2>&1; tail -15 gpurun_out/r04/pytest26.log
