#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
LINS_IESKF_LIB=$PWD/ab/prof2.so timeout 300 python tools/wave_phases.py 5 10 > gpurun_out/r04/wave_phases.txt 2>&1; cat gpurun_out/r04/wave_phases.txt
