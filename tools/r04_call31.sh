#!/bin/bash
# round 4, GPU call 31: the batch kernel under LLVM scheduler switches (relaxed occupancy, GCN trackers, no unclustered high-RP reschedule, metric bias 40, no post-RA scheduler)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 900 python tools/ab_timing.py ab/base.so ab/sc1.so ab/sc2.so ab/sc3.so ab/sc4.so ab/sc5.so mr > gpurun_out/r04/ab31.txt 2>&1; cat gpurun_out/r04/ab31.txt
