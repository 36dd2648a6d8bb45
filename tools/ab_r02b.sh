#!/bin/bash
# One GPU call of the second half of round 2: parity suite on the current build, the serial tail's building blocks in
# cycles, A/B kernel timing of the saved builds under ab/ against the current one.  Output under gpurun_out/ab/.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/ab
export LINS_ENABLE_DEBUG_KNOBS=1
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/ab/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/ab/pytest.log
tail -3 gpurun_out/ab/pytest.log
LINS_IESKF_LIB=$PWD/ab/fold.so timeout 300 python -m pytest tests/test_gpu_split.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > gpurun_out/ab/pytest_fold.log 2>&1; echo "pytest rc $?" >> gpurun_out/ab/pytest_fold.log
tail -3 gpurun_out/ab/pytest_fold.log
timeout 120 python tools/tail_cycles.py 1 > gpurun_out/ab/tail_cycles.txt 2>&1
cat gpurun_out/ab/tail_cycles.txt
cp "lins---lidar-inertial-slam_amd/liblins_ieskf.so" ab/cur.so
timeout 400 python tools/ab_timing.py ab/*.so mr > gpurun_out/ab/ab_timing.txt 2>&1
cat gpurun_out/ab/ab_timing.txt
