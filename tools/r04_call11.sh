#!/bin/bash
# round 4, GPU call 11: phase profile of the feature front-end (LINS_FE_PROF build: shader-clock ticks per phase, scan 0)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
LINS_IESKF_LIB=$PWD/ab/feprof.so timeout 300 python tools/frontend_rate.py 256 > gpurun_out/r04/fe_prof.txt 2>&1; tail -5 gpurun_out/r04/fe_prof.txt
timeout 300 python tools/frontend_rate.py 256 2>&1 | tail -1
