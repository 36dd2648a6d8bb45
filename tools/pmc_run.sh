#!/bin/bash
# usage: tools/pmc_run.sh <tag> <bench args...> ; runs kernel-trace + several PMC passes (own runs each,
# never combined with other trace domains) and writes gpurun_out/<tag>.*.txt summaries
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  rm -rf $root/gpurun_out/${tag}_$name
  rocprofv3 --kernel-trace --pmc "$@" -d $root/gpurun_out/${tag}_$name -- python $root/bench.py --steps 2 --warmup 1 --no-cpu --no-extras $BENCH_ARGS > $root/gpurun_out/${tag}_$name.log 2>&1
  python $root/tools/rocpd_summary.py $(find $root/gpurun_out/${tag}_$name -name "*.db") | grep -E "ieskf|^kernel" | sed 's/void lins:://; s/([^)]*)//' > $root/gpurun_out/${tag}.$name.txt
  rm -rf $root/gpurun_out/${tag}_$name
}
BENCH_ARGS="$*"
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cat $root/gpurun_out/${tag}.*.txt
