#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q --maxfail=12 > gpurun_out/r04/pytest4.log 2>&1
tail -5 gpurun_out/r04/pytest4.log
LINS_IESKF_LIB=$PWD/ab/prof2.so timeout 300 python tools/wave_phases.py 5 10 > gpurun_out/r04/wave_phases.txt 2>&1; cat gpurun_out/r04/wave_phases.txt
LINS_IESKF_LIB=$PWD/ab/prof2.so timeout 300 python tools/wave_phases.py 5 8 > gpurun_out/r04/wave_phases_8.txt 2>&1; tail -12 gpurun_out/r04/wave_phases_8.txt
