#!/bin/bash
# round 4, GPU call 10: the walk's windows — one scan per round, and the whole window at once when it holds few points (LINS_ONESHOT_PTS)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python tools/ab_timing.py ab/base.so ab/os96.so ab/os256.so ab/osinf.so mr > gpurun_out/r04/ab10b.txt 2>&1; cat gpurun_out/r04/ab10.txt
for v in os96; do
  LINS_IESKF_LIB=$PWD/ab/$v.so timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest10b_$v.log 2>&1; echo "$v: $(tail -1 gpurun_out/r04/pytest10b_$v.log)"
done
