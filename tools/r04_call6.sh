#!/bin/bash
# round 4, GPU call 6: walk cache (the second / third points of a query's previous nearest neighbour) — suite, A/B, wave profile
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r04/pytest6.log 2>&1
tail -15 gpurun_out/r04/pytest6.log
timeout 900 python tools/ab_timing.py ab/prev.so ab/cache.so ab/nocache.so mr > gpurun_out/r04/ab6.txt 2>&1
cat gpurun_out/r04/ab6.txt
LINS_IESKF_LIB=$PWD/ab/prof2.so timeout 300 python tools/wave_phases.py 5 10 > gpurun_out/r04/wave_phases_cache.txt 2>&1; cat gpurun_out/r04/wave_phases_cache.txt
