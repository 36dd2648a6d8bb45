#!/bin/bash
# round 4, GPU call 19: certificate margins re-swept on the round-3/4 scans (round 2 chose 0.08 / 0.035 on the lighter scans)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python tools/margin_sweep.py 0.15,0.20,0.27,0.35,0.50 0.06,0.08,0.11,0.15,0.20 > gpurun_out/r04/margin_sweep.txt 2>&1; cat gpurun_out/r04/margin_sweep.txt
