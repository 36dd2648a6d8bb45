#!/bin/bash
# round 4, GPU call 7: do the three "guarded by a flag / by noinline" code-generation sensitivities of round 3 still reproduce?
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
LINS_IESKF_LIB=$PWD/ab/gridinl.so timeout 600 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r04/pytest_gridinl.log 2>&1; tail -8 gpurun_out/r04/pytest_gridinl.log
LINS_IESKF_LIB=$PWD/ab/slpon.so timeout 600 python -m pytest tests/test_gpu_map.py tests/test_gpu_edge_cases.py tests/test_gpu_ref.py -m gpu -q --maxfail=30 > gpurun_out/r04/pytest_slpon.log 2>&1; tail -12 gpurun_out/r04/pytest_slpon.log
