#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_sequence.py -m gpu -q --maxfail=10 -k "streams or one_kernel or sequence or chain" > gpurun_out/r04/pytest25.log 2>&1; tail -2 gpurun_out/r04/pytest25.log
LINS_ENABLE_DEBUG_KNOBS=1 LINS_LAUNCH_ORDER=0 timeout 300 python tools/streams_rate.py 1024 2>&1 | tail -2 | cut -c1-200 | tee gpurun_out/r04/streams25.txt
timeout 300 python tools/streams_rate.py 1024 2>&1 | tail -2 | cut -c1-200 | tee -a gpurun_out/r04/streams25.txt
