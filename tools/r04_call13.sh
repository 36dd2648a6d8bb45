#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
for v in fec0 fec64 fec0 fec64; do echo -n "$v: "; LINS_IESKF_LIB=$PWD/ab/$v.so timeout 300 python tools/frontend_rate.py 256 2>&1 | tail -1 | cut -c1-110; done | tee gpurun_out/r04/fe_ab13.txt
