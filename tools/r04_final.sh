#!/bin/bash
# round 4, final check of the committed sources on the GPU box: build stamp, smoke(), the whole GPU suite, the default bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > gpurun_out/r04/final_smoke.log 2>&1; tail -2 gpurun_out/r04/final_smoke.log
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r04/final_pytest_gpu.log 2>&1; tail -2 gpurun_out/r04/final_pytest_gpu.log
( time python bench.py ) > gpurun_out/r04/final_bench.json 2> gpurun_out/r04/final_bench.err; tail -4 gpurun_out/r04/final_bench.err; tail -1 gpurun_out/r04/final_bench.json | cut -c1-260
