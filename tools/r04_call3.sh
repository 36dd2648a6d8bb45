#!/bin/bash
# round 4, GPU call 3: dense layout from iteration k on (query <-> lane change through the hand-over words), A/B against the previous commit
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r04/pytest3.log 2>&1
tail -30 gpurun_out/r04/pytest3.log
timeout 900 python tools/ab_timing.py ab/prev.so ab/dense.so ab/dense.so:LINS_DENSE_FROM=0 ab/dense.so:LINS_DENSE_FROM=1 ab/dense.so:LINS_DENSE_FROM=2 ab/dense.so:LINS_DENSE_FROM=3 \
  ab/dense.so:LINS_DENSE_FROM=6 ab/dense.so:LINS_DENSE_FROM=4,LINS_RELAY_AT=0 ab/dense.so:LINS_DENSE_FROM=4,LINS_TAIL_AT=4 ab/dense.so:LINS_DENSE_FROM=2,LINS_RELAY_AT=2 mr > gpurun_out/r04/ab3.txt 2>&1
cat gpurun_out/r04/ab3.txt
