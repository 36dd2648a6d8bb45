#!/bin/bash
# The line-of-record artifacts that follow a change of the feature front-end only (a subset of tools/refresh_profiles.sh:
# the batch kernel's anatomy, PMC and parity sweeps keep their records — its object file is unchanged — but the library's
# source digest moves, so the traffic record bench.py reads is taken again).  usage (GPU box): tools/refresh_frontend.sh r05
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
rm -f $out/canaries.log
timeout 900 python -m pytest tests -q -m gpu > $out/${tag}_pytest_gpu.log 2>&1; tail -2 $out/${tag}_pytest_gpu.log
cat $out/canaries.log >> $out/${tag}_pytest_gpu.log 2>/dev/null
timeout 600 python tools/pmc_traffic.py $tag > $out/${tag}_pmc_traffic.log 2>&1; tail -3 $out/${tag}_pmc_traffic.log
cp $out/${tag}_pmc_traffic.json profiles/ 2>/dev/null   # (bench.py reads the record from profiles/, stamp-checked)
timeout 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json | cut -c1-300
{
  timeout 300 python tools/e2e_rate.py 2>&1 | tail -2
  timeout 300 python tools/streams_rate.py 1024 2>&1 | tail -2
  timeout 300 python tools/map_rate.py 2>&1 | tail -4
  timeout 300 python tools/reproject_rate.py 2>&1 | tail -1
  timeout 300 python tools/frontend_rate.py 2>&1 | tail -2
  timeout 300 python tools/frontend_rate.py 1024 2>&1 | tail -1
} > $out/${tag}_aux_rates.txt 2>&1
cat $out/${tag}_aux_rates.txt
timeout 600 bash tools/aux_profiles.sh > /dev/null 2>&1; cp $out/aux_kernel_stats.csv $out/${tag}_rocprofv3_kernel_stats_aux.csv; cat $out/${tag}_rocprofv3_kernel_stats_aux.csv
timeout 600 bash tools/aux_pmc.sh > /dev/null 2>&1; cp $out/aux_pmc.txt $out/${tag}_rocprofv3_pmc_aux.txt; cat $out/${tag}_rocprofv3_pmc_aux.txt
timeout 600 python tools/frontend_vs_libm.py 256 --gpu > $out/${tag}_frontend_vs_libm.txt 2>&1; tail -12 $out/${tag}_frontend_vs_libm.txt
timeout 600 python tools/frontend_sweep.py 1024 50000 > $out/${tag}_frontend_sweep.txt 2>&1; tail -2 $out/${tag}_frontend_sweep.txt
timeout 600 python tools/frontend_sweep.py 512 90000 open >> $out/${tag}_frontend_sweep.txt 2>&1; tail -2 $out/${tag}_frontend_sweep.txt
