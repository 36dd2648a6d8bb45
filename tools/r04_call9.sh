#!/bin/bash
# round 4, GPU call 9: the upper rings' scans batched (LINS_GLOB_BATCH), queries dealt to the waves (LINS_INTERLEAVE), the solve chain at
# issue priority (LINS_SOLVE_PRIO), sector range pruning (LINS_RANGE_PRUNE): kernel time of each build, one call; parity of two of them.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python tools/ab_timing.py ab/base.so ab/gb4.so ab/gb4i.so ab/gb4ip.so ab/rp.so mr > gpurun_out/r04/ab9.txt 2>&1; cat gpurun_out/r04/ab9.txt
for v in gb4ip rp; do
  LINS_IESKF_LIB=$PWD/ab/$v.so timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest9_$v.log 2>&1; echo "$v: $(tail -1 gpurun_out/r04/pytest9_$v.log)"
done
