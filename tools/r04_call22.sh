#!/bin/bash
# round 4, GPU call 22: the reach tests of a search issued unconditionally (ab_a), + the scans' insertion without a branch (ab_both)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python tools/ab_timing.py ab/base.so ab/ab_a.so ab/ab_both.so mr > gpurun_out/r04/ab22.txt 2>&1; cat gpurun_out/r04/ab22.txt
LINS_IESKF_LIB=$PWD/ab/ab_both.so timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest22.log 2>&1; tail -1 gpurun_out/r04/pytest22.log
