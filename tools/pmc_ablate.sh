#!/bin/bash
# SQ instruction counters of the LDS kernel with the walk / the whole search skipped (profiling aid)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for f in 0 1 2; do
  rm -rf $root/gpurun_out/abl_$f
  LINS_DEBUG_SKIP=$f rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $root/gpurun_out/abl_$f -- python $root/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
  echo "== skip=$f"
  python $root/tools/rocpd_summary.py $(find $root/gpurun_out/abl_$f -name "*.db") | grep ieskf_lds | sed 's/ieskf_lds_kernel<[^>]*>//; s/void lins:://; s/([^)]*)//' | cut -c1-80
  rm -rf $root/gpurun_out/abl_$f
done
