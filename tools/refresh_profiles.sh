#!/bin/bash
# Regenerates the line-of-record artifacts for round $1 (default r05) on the GPU box into gpurun_out/; copy what is
# to be judged into profiles/ afterwards (tools/README.md).  PMC passes are separate rocprofv3 runs with
# --kernel-trace only (never combined with other trace domains).
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
rm -f $out/canaries.log
python -m pytest tests -q -m gpu > $out/${tag}_pytest_gpu.log 2>&1; tail -2 $out/${tag}_pytest_gpu.log
cat $out/canaries.log >> $out/${tag}_pytest_gpu.log 2>/dev/null
python tools/pmc_traffic.py $tag > $out/${tag}_pmc_traffic.log 2>&1; tail -3 $out/${tag}_pmc_traffic.log
cp $out/${tag}_pmc_traffic.json profiles/ 2>/dev/null   # (bench.py reads the record from profiles/, stamp-checked)
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json | cut -c1-300
# kernel trace of the bench command proper (warm-up + timed steps only: the batch kernel's average IS the step's kernel
# time) and of the command with its extra blocks (30-iteration launches of the stop-rule block, the single-scan kernel)
( cd /tmp && export TMPDIR=/tmp && rm -rf $out/${tag}_kt && rocprofv3 --kernel-trace --stats -d $out/${tag}_kt -- python $root/bench.py --no-cpu --no-extras > $out/${tag}_kt.log 2>&1
  python $root/tools/rocpd_summary.py $(find $out/${tag}_kt -name "*.db") > $out/${tag}_kernel_stats.csv; rm -rf $out/${tag}_kt
  rocprofv3 --kernel-trace --stats -d $out/${tag}_kt -- python $root/bench.py --no-cpu > $out/${tag}_kt_extras.log 2>&1
  python $root/tools/rocpd_summary.py $(find $out/${tag}_kt -name "*.db") > $out/${tag}_kernel_stats_extras.csv; rm -rf $out/${tag}_kt )
cat $out/${tag}_kernel_stats.csv $out/${tag}_kernel_stats_extras.csv
{
  python tools/e2e_rate.py 2>&1 | tail -2
  python tools/streams_rate.py 1024 2>&1 | tail -2
  python tools/map_rate.py 2>&1 | tail -4
  python tools/reproject_rate.py 2>&1 | tail -1
  python tools/frontend_rate.py 2>&1 | tail -2
} > $out/${tag}_aux_rates.txt 2>&1
cat $out/${tag}_aux_rates.txt
{
  echo "== tools/late_iter_time.py mr  (one late iteration of the batch kernel, by phase)"
  python tools/late_iter_time.py mr 2>&1 | tail -8
  echo "== tools/iter_curve.py mr 10"
  python tools/iter_curve.py mr 10 2>&1 | tail -10
  echo "== tools/cold_iter_time.py  (set-up and the cold iteration)"
  python tools/cold_iter_time.py 2>&1 | tail -7
  echo "== tools/wg_cost_model.py 1024 mr  (tail of the launch)"
  python tools/wg_cost_model.py 1024 mr 2>&1 | tail -3
  echo "== tools/step_modes.py  (host sync per step / back to back / pipelined gather mode)"
  python tools/step_modes.py 2>&1 | tail -3
  echo "== tools/relay_sweep.py 1024 0 3 4 5 6  (iterations per part of the several-part updates; 0 = whole updates)"
  python tools/relay_sweep.py 1024 0 3 4 5 6 2>&1 | tail -5
  echo "== RS_ITERS=30 RS_FIXED=0 tools/relay_sweep.py 1024 0 4  (the reference's stop rule)"
  RS_ITERS=30 RS_FIXED=0 python tools/relay_sweep.py 1024 0 4 2>&1 | tail -2
  echo "== tools/stop_rule_timeline.py  (whole updates under the stop rule: who ends the launch)"
  python tools/stop_rule_timeline.py 2>&1 | tail -9
  if [ -f ab/qtrace.so ]; then
    echo "== LINS_IESKF_LIB=ab/qtrace.so tools/queue_trace.py 1024 10 1 / 1024 30 0  (the ticketed launch's timeline; -DLINS_QUEUE_TRACE=1 build)"
    LINS_IESKF_LIB=$PWD/ab/qtrace.so python tools/queue_trace.py 1024 10 1 2>&1 | tail -9
    LINS_IESKF_LIB=$PWD/ab/qtrace.so python tools/queue_trace.py 1024 30 0 2>&1 | tail -9
  fi
  echo "== tools/split_launch_time.py  (one launch per run against two launch queues; queued runs and one run + wait)"
  python tools/split_launch_time.py 2>&1 | tail -4
  echo "== tools/slow_scans.py 1024 mr  (the slowest updates of the batch, PROF variant, whole updates)"
  python tools/slow_scans.py 1024 mr 2>&1 | tail -13
  echo "== tools/index_time.py 1024  (grid_index_kernel at lins_batch_upload)"
  python tools/index_time.py 1024 2>&1 | tail -1
  if [ -f ab/prof2.so ]; then
    echo "== LINS_IESKF_LIB=ab/prof2.so tools/wave_phases.py 5 10  (per-wave phases of the late iterations; -DLINS_PROF2=5 build)"
    LINS_IESKF_LIB=$PWD/ab/prof2.so python tools/wave_phases.py 5 10 2>&1 | head -14
  fi
  if [ -f ab/prof0.so ]; then
    echo "== LINS_IESKF_LIB=ab/prof0.so tools/wave_phases.py 0 1 / 0 4  (the cold iteration; iterations 0-3; -DLINS_PROF2=0 build)"
    LINS_IESKF_LIB=$PWD/ab/prof0.so python tools/wave_phases.py 0 1 2>&1 | head -11
    LINS_IESKF_LIB=$PWD/ab/prof0.so python tools/wave_phases.py 0 4 2>&1 | head -11
  fi
} > $out/${tag}_kernel_anatomy.txt 2>&1
cat $out/${tag}_kernel_anatomy.txt
bash tools/aux_profiles.sh > /dev/null 2>&1; cp $out/aux_kernel_stats.csv $out/${tag}_rocprofv3_kernel_stats_aux.csv; cat $out/${tag}_rocprofv3_kernel_stats_aux.csv
bash tools/aux_pmc.sh > /dev/null 2>&1; cp $out/aux_pmc.txt $out/${tag}_rocprofv3_pmc_aux.txt; cat $out/${tag}_rocprofv3_pmc_aux.txt
python tools/frontend_vs_libm.py 256 --gpu > $out/${tag}_frontend_vs_libm.txt 2>&1; tail -12 $out/${tag}_frontend_vs_libm.txt
bash tools/pmc_run.sh ${tag}pmc --batch 1024 --search auto > $out/${tag}_pmc_all.txt 2>&1
tail -40 $out/${tag}_pmc_all.txt
{
  python tools/parity_sweep.py 2048 60000 2>&1 | tail -7
  python tools/parity_sweep.py 1024 70000 wide 2>&1 | tail -7
  python tools/parity_sweep.py 2048 80000 open 2>&1 | tail -7
  python tools/frontend_sweep.py 1024 50000 2>&1 | tail -2
} > $out/${tag}_parity_sweep.txt 2>&1
cat $out/${tag}_parity_sweep.txt
