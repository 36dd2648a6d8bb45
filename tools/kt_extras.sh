#!/bin/bash
# rocprofv3 kernel trace of `bench.py --no-cpu` WITH its extra blocks (stop-rule launches, the single-scan kernel).
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out; mkdir -p "$out"
d=$(mktemp -d /tmp/kt_XXXXXX)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$d" -- python "$root/bench.py" --no-cpu > "$out/r02_kt_extras.log" 2>&1
python "$root/tools/rocpd_summary.py" $(find "$d" -name "*.db") > "$out/r02_kernel_stats_extras.csv"
cut -c1-120 "$out/r02_kernel_stats_extras.csv"
