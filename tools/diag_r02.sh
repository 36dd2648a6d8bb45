#!/bin/bash
# Round-2 diagnostic pass over the dominant kernel (run on the GPU box): baseline bench line, phase
# profile, the SQ / instruction-cache counters that say WHY waves wait, and a PC-sampling attempt.
# Everything lands in gpurun_out/diag/.  Each step is bounded by its own timeout.
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/diag
mkdir -p $out
cd $root
timeout 300 python bench.py --steps 20 > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
timeout 120 python tools/phase_profile.py 1024 mr > $out/phase.txt 2>&1; cat $out/phase.txt
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 -L > $out/counters.txt 2>&1
grep -i -E "icache|ifetch|INST_CACHE|SQC_" $out/counters.txt | head -40
run() { # name counters...
  name=$1; shift
  rm -rf $out/pmc_$name
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d $out/pmc_$name -- python $root/bench.py --steps 2 --warmup 1 --no-cpu > $out/pmc_$name.log 2>&1
  python $root/tools/rocpd_summary.py $(find $out/pmc_$name -name "*.db") | grep -E "ieskf|^kernel" | sed 's/void lins:://; s/([^)]*)//' > $out/pmc.$name.txt
  rm -rf $out/pmc_$name
  cat $out/pmc.$name.txt
}
run ic SQ_IFETCH SQ_WAIT_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run st SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT
run dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
# PC sampling (beta): stochastic first, host_trap as the fallback
for m in stochastic host_trap; do
  rm -rf $out/pcs_$m
  if [ $m = stochastic ]; then unit=cycles; iv=65536; else unit=time; iv=1; fi
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m --pc-sampling-unit $unit --pc-sampling-interval $iv \
     --kernel-trace --output-format csv -d $out/pcs_$m -- python $root/bench.py --steps 30 --warmup 1 --no-cpu > $out/pcs_$m.log 2>&1
  echo "pc sampling $m rc=$?"; tail -3 $out/pcs_$m.log
  find $out/pcs_$m -type f | head; du -sh $out/pcs_$m
  f=$(find $out/pcs_$m -name "*pc_sampling*csv" | head -1)
  if [ -n "$f" ]; then
    head -3 $f
    python $root/tools/pcs_reduce.py $f > $out/pcs_$m.hist.txt 2>&1
    head -5 $out/pcs_$m.hist.txt
    rm -rf $out/pcs_$m
    break
  fi
  rm -rf $out/pcs_$m
done
