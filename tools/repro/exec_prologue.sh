#!/bin/bash
# Reproduces, without a GPU, the register-allocator fault of ROCm 7.2 clang that tools/check_exec_prologue.py describes
# (README.md in this directory, row `load_lds_grid`): compiles ieskf_lds.hip from two revisions of THIS repository's
# history whose GPU outcome is known and runs the checker on the assembly.
#   4be5773  round 6, lane merges of the three-lane block under uniform control flow (second revision), inlined grid load:
#            test_correspondences_bit_exact[lds], test_golden[lds] FAIL on the MI355X      -> 1 site (.LBB9_604)
#            ... the same source with -DLINS_GRID_INLINE=0: 207 GPU tests pass               -> 0 sites
#   182fd04  round 4 with -DLINS_GRID_INLINE=1 and round 4's flags: 23 GPU tests failed     -> 1 site (.LBB9_603, the same place)
#            ... as shipped in round 4 (-DLINS_GRID_INLINE=0): green                         -> 0 sites
# usage: tools/repro/exec_prologue.sh        (needs hipcc; ~1.5 min)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
pkg=lins---lidar-inertial-slam_amd
tmp=$(mktemp -d)
base="-O3 -Wall -Wno-unused-function --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing --offload-device-only -S"
build() {  # revision, output name, extra flags
  mkdir -p $tmp/$1 && (cd $root && git archive $1 $pkg/csrc include) | tar -x -C $tmp/$1
  (cd $tmp/$1/$pkg/csrc && /opt/rocm/bin/hipcc $base $3 -o $tmp/$2.s ieskf_lds.hip 2> /dev/null)
}
build 4be5773 r06_uniform_inlined "-mllvm -disable-machine-licm" &
build 4be5773 r06_uniform_out_of_line "-mllvm -disable-machine-licm -DLINS_GRID_INLINE=0" &
wait
build 182fd04 r04_inlined "-DLINS_GRID_INLINE=1" &
build 182fd04 r04_out_of_line "-DLINS_GRID_INLINE=0" &
wait
python $root/tools/check_exec_prologue.py $tmp/r06_uniform_inlined.s $tmp/r06_uniform_out_of_line.s $tmp/r04_inlined.s $tmp/r04_out_of_line.s | cut -c1-260 || true
rm -rf $tmp
