#!/bin/bash
# round 4, GPU call 2: suite on the committed build, relay-part sweep with the packed hand-over, tail variants, PMC traffic, late-iteration anatomy
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r04/pytest2.log 2>&1
tail -5 gpurun_out/r04/pytest2.log
timeout 900 python tools/ab_timing.py ab/orig.so ab/new.so ab/new.so:LINS_RELAY_AT=2 ab/new.so:LINS_RELAY_AT=3 ab/new.so:LINS_RELAY_AT=5 ab/new.so:LINS_RELAY_AT=6 \
  ab/s1o0.so:LINS_TAIL_AT=4 ab/s0o0.so:LINS_TAIL_AT=4 ab/s1o1.so:LINS_TAIL_AT=4 ab/s1o0.so:LINS_TAIL_AT=4,LINS_TAIL_DENSE=1 ab/s1o0.so:LINS_TAIL_AT=6,LINS_RELAY_AT=3,LINS_TAIL_DENSE=1 mr > gpurun_out/r04/ab2.txt 2>&1
cat gpurun_out/r04/ab2.txt
timeout 600 python tools/pmc_traffic.py r04a > gpurun_out/r04/pmc_traffic.log 2>&1; tail -5 gpurun_out/r04/pmc_traffic.log
LINS_TAIL_AT=0 timeout 600 python tools/late_iter_time.py mr > gpurun_out/r04/late_iter_head.txt 2>&1; cat gpurun_out/r04/late_iter_head.txt
