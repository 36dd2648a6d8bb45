#!/bin/bash
# round 4, GPU call 1: the whole GPU suite on the new library, then A/B kernel timing of the cut variants against round 3's build
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r04/pytest1.log 2>&1
tail -30 gpurun_out/r04/pytest1.log
timeout 700 python tools/ab_timing.py ab/orig.so ab/new.so ab/new.so:LINS_TAIL_AT=0 ab/new.so:LINS_TAIL_DENSE=1 \
  ab/new.so:LINS_TAIL_AT=4,LINS_RELAY_AT=2 ab/new.so:LINS_TAIL_AT=2,LINS_RELAY_AT=0 ab/new.so:LINS_TAIL_AT=6,LINS_RELAY_AT=3 \
  ab/new.so:LINS_TAIL_AT=3,LINS_RELAY_AT=0,LINS_TAIL_DENSE=1 mr > gpurun_out/r04/ab1.txt 2>&1
cat gpurun_out/r04/ab1.txt
