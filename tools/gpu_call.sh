#!/bin/bash
# ONE GPU call made of named steps (replaces the one-off r04_call*.sh scripts of round 4, which are in the history at
# commit 182fd04).  Every step logs to gpurun_out/<tag>/<step>.log and prints its last lines.
#   usage: tools/gpu_call.sh <tag> <step> [<step> ...]            e.g.
#          /usr/local/graft/bin/gpurun --timeout 900 -- 'tools/gpu_call.sh r05a smoke pytest bench'
#   steps  smoke                      __graft_entry__.build() + smoke()
#          pytest[=<pytest args>]     the GPU suite (default: tests -m gpu -q), e.g. "pytest=tests/test_gpu_parity.py -m gpu -q -k 'cut or walk'" (eval'd: inner quotes work)
#          libtest=<variant>[,<pytest args>]   the same against ab/<variant>.so (tools/build_variant.sh), LINS_IESKF_LIB
#          bench[=<bench.py args>]    the bench line (default flags)
#          ab[=<mode>]                tools/ab_timing.py over every ab/*.so + the in-tree library ("mr": batch kernel time)
#          py=<script and args>       any tools/*.py, e.g. 'py=tools/relay_sweep.py 1024 0 4'
#          sh=<script and args>       any tools/*.sh
#          set=<VAR>=<value>          export an environment variable for the steps that follow (set=<VAR>= clears it)
#   LINS_ENABLE_DEBUG_KNOBS=1 is exported (the tools' knobs); a step that fails does not stop the call.
cd "$(dirname "$0")/.." || exit 1
tag=$1; shift
out=gpurun_out/$tag; mkdir -p "$out"
export LINS_ENABLE_DEBUG_KNOBS=1
k=0
for step in "$@"; do
  k=$((k + 1)); name=${step%%=*}; arg=""; [[ "$step" == *=* ]] && arg=${step#*=}
  log="$out/$(printf %02d $k)_$name.log"
  case $name in
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > "$log" 2>&1 ;;
    pytest) eval "timeout 1200 python -m pytest ${arg:-tests -m gpu -q}" > "$log" 2>&1 ;;
    libtest) v=${arg%%,*}; rest=""; [[ "$arg" == *,* ]] && rest=${arg#*,}
             LINS_IESKF_LIB=$PWD/ab/$v.so eval "timeout 1200 python -m pytest ${rest:-tests -m gpu -q}" > "$log" 2>&1 ;;
    bench) ( time timeout 900 python bench.py $arg ) > "$out/$(printf %02d $k)_bench.json" 2> "$log"; tail -1 "$out/$(printf %02d $k)_bench.json" | cut -c1-400 ;;
    ab) timeout 900 python tools/ab_timing.py ab/*.so lins---lidar-inertial-slam_amd/liblins_ieskf.so ${arg:-mr} > "$log" 2>&1 ;;
    py) timeout 900 python $arg > "$log" 2>&1 ;;
    sh) timeout 1500 bash $arg > "$log" 2>&1 ;;
    set) export "$arg"; echo "exported $arg" > "$log" ;;
    *) echo "unknown step $step" > "$log" ;;
  esac
  echo "== [$k] $step (rc $?)"; tail -${GPU_CALL_TAIL:-6} "$log"
done
