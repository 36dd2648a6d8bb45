#!/bin/bash
# extra SQ counters (instruction fetch, LDS wait, scratch/VMEM) for the bench kernel; one block per run
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rm -rf $root/gpurun_out/x_$name
  rocprofv3 --kernel-trace --pmc "$@" -d $root/gpurun_out/x_$name -- python $root/bench.py --steps 2 --warmup 1 --no-cpu $BENCH_ARGS > $root/gpurun_out/x_$name.log 2>&1
  python $root/tools/rocpd_summary.py $(find $root/gpurun_out/x_$name -name "*.db") | grep -E "ieskf_lds" | sed 's/void lins:://; s/([^)]*)//'
  rm -rf $root/gpurun_out/x_$name
}
BENCH_ARGS="$*"
run a SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD
run c SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_CYCLES SQ_BUSY_CU_CYCLES
