#!/bin/bash
# Instruction mix of ONE late iteration of the persistent kernel: counters of runs with 7 and 10 fixed iterations
# (own PMC passes, kernel-trace only); the slope between them is the cost of an iteration in which nearly every
# certificate holds.  With LINS_DEBUG_SKIP counting aids the slope is split by phase:
#   0x10000 no solve / update, 0x20000 no rows, 0x40000 no row reduction, 3 no searches.
# usage: tools/late_iter_pmc.sh [search]   -> gpurun_out/late_iter_pmc.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
search=${1:-mr}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
export LINS_ENABLE_DEBUG_KNOBS=1
out=$root/gpurun_out/late_iter_pmc.txt
: > $out
for skip in ${SKIPS:-0 65536 196608 458752 458755}; do
  export LINS_DEBUG_SKIP=$skip
  for it in 7 10; do
    d=$root/gpurun_out/_lip_$it
    rm -rf $d
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $d -- python $root/tools/iters_run.py $it $search > /dev/null 2>&1
    echo "== skip $skip iterations $it" >> $out
    python $root/tools/rocpd_summary.py $(find $d -name "*.db") | grep -E "ieskf_lds" | sed 's/void lins:://; s/([^)]*)//; s/lds_mr::ieskf_lds_kernel<512, 1, false, false, false, false>,//' >> $out
    rm -rf $d
  done
done
python - <<PY
import re
rows = {}
cur = None
for line in open("$out"):
    m = re.match(r"== skip (\d+) iterations (\d+)", line)
    if m:
        cur = (int(m.group(1)), int(m.group(2))); rows[cur] = {}
        continue
    f = line.strip().split(",")
    if len(f) == 3 and f[0].startswith("SQ_"):
        rows[cur][f[0]] = float(f[1])
    elif len(f) > 4 and f[0].isdigit():
        rows[cur]["avg_ns"] = float(f[2])
print("per late iteration and wave (1024 workgroups x 8 waves, 3 launches):")
for skip in sorted({k[0] for k in rows}):
    a, b = rows[(skip, 7)], rows[(skip, 10)]
    d = {k: (b[k] - a[k]) / 9 for k in b if k in a}
    w = 1024 * 8
    print(f"skip {skip:#x}: time {(b['avg_ns'] - a['avg_ns']) / 3 / 1e3:.1f} us/iteration; per wave: VALU {d['SQ_INSTS_VALU'] / w:.0f} SALU {d['SQ_INSTS_SALU'] / w:.0f} "
          f"LDS {d['SQ_INSTS_LDS'] / w:.0f} VMEM {d['SQ_INSTS_VMEM'] / w:.0f}; active lanes per VALU instruction {d['SQ_THREAD_CYCLES_VALU'] / d['SQ_INSTS_VALU'] / 4:.1f}")
PY
