#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
for it in 1 4; do LINS_IESKF_LIB=$PWD/ab/prof0c.so timeout 300 python tools/wave_counts.py $it 2>&1 | tail -13; done | tee gpurun_out/r04/wave_counts.txt
