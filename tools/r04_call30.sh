#!/bin/bash
# round 4, GPU call 29: the front-end's pick rounds of four rings in one wave (LINS_FE_ROWS): rate against the build before, parity
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
for v in fe_one fe_two fe_one fe_two; do echo -n "$v: "; LINS_IESKF_LIB=$PWD/ab/$v.so timeout 300 python tools/frontend_rate.py 256 2>&1 | tail -1 | cut -c1-110; done | tee gpurun_out/r04/fe_ab30.txt
LINS_IESKF_LIB=$PWD/ab/fe_two.so timeout 600 python -m pytest tests/test_frontend_oracle.py tests/test_gpu_edge_cases.py tests/test_gpu_ref.py tests/test_gpu_sequence.py -m gpu -q --maxfail=10 > gpurun_out/r04/pytest30.log 2>&1; tail -3 gpurun_out/r04/pytest30.log
