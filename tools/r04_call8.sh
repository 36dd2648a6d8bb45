#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
for it in 1 2 4; do
  echo "===== iterations 0..$((it-1))"
  LINS_IESKF_LIB=$PWD/ab/prof0.so timeout 300 python tools/wave_phases.py 0 $it 2>&1 | head -16
done > gpurun_out/r04/wave_phases_early.txt 2>&1
cat gpurun_out/r04/wave_phases_early.txt
