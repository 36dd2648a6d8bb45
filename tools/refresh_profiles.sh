#!/bin/bash
# Regenerates the line-of-record artifacts for round $1 (default r01) on the GPU box into gpurun_out/.
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json
python tools/e2e_rate.py > $out/${tag}_e2e.txt 2>&1; cat $out/${tag}_e2e.txt
python tools/reproject_rate.py > $out/${tag}_reproject.txt 2>&1; cat $out/${tag}_reproject.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf $out/${tag}_kt && rocprofv3 --kernel-trace --stats -d $out/${tag}_kt -- python $root/bench.py --no-cpu --no-extras > $out/${tag}_kt.log 2>&1
  python $root/tools/rocpd_summary.py $(find $out/${tag}_kt -name "*.db") > $out/${tag}_kernel_stats.csv; rm -rf $out/${tag}_kt )
cat $out/${tag}_kernel_stats.csv
bash tools/pmc_run.sh ${tag}pmc --batch 1024 --search auto > $out/${tag}_pmc_all.txt 2>&1
cat $out/${tag}_pmc_all.txt
