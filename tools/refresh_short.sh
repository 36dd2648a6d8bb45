#!/bin/bash
# The line-of-record artifacts that follow a change of the batch kernel (a subset of tools/refresh_profiles.sh: the rows
# around the update keep their records).  usage (GPU box): tools/refresh_short.sh r03b  -> gpurun_out/r03b_*
tag=${1:-r03b}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python tools/pmc_traffic.py $tag > $out/${tag}_pmc_traffic.log 2>&1; tail -3 $out/${tag}_pmc_traffic.log
cp $out/${tag}_pmc_traffic.json profiles/ 2>/dev/null   # (bench.py reads the record from profiles/, stamp-checked)
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json | cut -c1-300
( cd /tmp && export TMPDIR=/tmp && rm -rf $out/${tag}_kt && rocprofv3 --kernel-trace --stats -d $out/${tag}_kt -- python $root/bench.py --no-cpu --no-extras > $out/${tag}_kt.log 2>&1
  python $root/tools/rocpd_summary.py $(find $out/${tag}_kt -name "*.db") > $out/${tag}_kernel_stats.csv; rm -rf $out/${tag}_kt )
cat $out/${tag}_kernel_stats.csv
{
  echo "== tools/relay_sweep.py 1024 0 4 5 6 7 8  (iterations per part of the several-part updates; 0 = whole updates)"
  python tools/relay_sweep.py 1024 0 4 5 6 7 8 2>&1 | tail -6
  echo "== RS_ITERS=30 RS_FIXED=0 tools/relay_sweep.py 1024 0 6  (the reference's stop rule)"
  RS_ITERS=30 RS_FIXED=0 python tools/relay_sweep.py 1024 0 6 2>&1 | tail -2
  echo "== tools/relay_sweep.py 640 / 2048 / 4096 0 6"
  python tools/relay_sweep.py 640 0 6 2>&1 | tail -2
  python tools/relay_sweep.py 2048 0 6 2>&1 | tail -2
  python tools/relay_sweep.py 4096 0 6 2>&1 | tail -2
  echo "== tools/index_time.py 1024  (grid_index_kernel at lins_batch_upload)"
  python tools/index_time.py 1024 2>&1 | tail -1
  echo "== tools/phase_profile.py 1024 mr  (whole updates: the phase profile switches the relay off)"
  python tools/phase_profile.py 1024 mr 2>&1 | head -8
  echo "== tools/wg_cost_model.py 1024 mr  (tail of a launch of whole updates)"
  python tools/wg_cost_model.py 1024 mr 2>&1 | tail -3
  echo "== tools/step_modes.py"
  python tools/step_modes.py 2>&1 | tail -3
} > $out/${tag}_kernel_anatomy.txt 2>&1
cat $out/${tag}_kernel_anatomy.txt
{
  python tools/e2e_rate.py 2>&1 | tail -2
  python tools/streams_rate.py 1024 2>&1 | tail -2
} > $out/${tag}_aux_rates.txt 2>&1
cat $out/${tag}_aux_rates.txt
{
  python tools/parity_sweep.py 2048 60000 2>&1 | tail -7
  python tools/parity_sweep.py 1024 70000 wide 2>&1 | tail -7
} > $out/${tag}_parity_sweep.txt 2>&1
cat $out/${tag}_parity_sweep.txt
python -m pytest tests -q -m gpu > $out/${tag}_pytest_gpu.log 2>&1; tail -2 $out/${tag}_pytest_gpu.log
