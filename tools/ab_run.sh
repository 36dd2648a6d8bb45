#!/bin/bash
# A/B kernel timing of every ab/*.so in ONE GPU call + (optionally) the whole GPU suite against named variants.
# usage: tools/ab_run.sh [variant-to-test ...]   -> gpurun_out/ab/
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/ab
export LINS_ENABLE_DEBUG_KNOBS=1
for v in "$@"; do
  LINS_IESKF_LIB=$PWD/ab/$v.so timeout 400 python -m pytest tests -m gpu -q -x > gpurun_out/ab/pytest_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/ab/pytest_$v.log)"
done
timeout 600 python tools/ab_timing.py ab/*.so mr > gpurun_out/ab/ab_timing.txt 2>&1
cat gpurun_out/ab/ab_timing.txt
