#!/bin/bash
# rocprofv3 kernel trace of the bench command proper (`bench.py --no-cpu --no-extras`: warm-up + timed launches only).
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out; mkdir -p "$out"
d=$(mktemp -d /tmp/kt_XXXXXX)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$d" -- python "$root/bench.py" --no-cpu --no-extras > "$out/r02_kt.log" 2>&1
python "$root/tools/rocpd_summary.py" $(find "$d" -name "*.db") > "$out/r02_kernel_stats.csv"
cut -c1-100 "$out/r02_kernel_stats.csv"; grep -o '"kernel_ms": [0-9.]*' "$out/r02_kt.log" | head -1
