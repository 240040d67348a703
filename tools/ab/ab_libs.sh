#!/bin/bash
# A/B/C... on one GPU box: the same bench lines through several builds of the library, alternating, so that box-to-box
# clock differences cancel.  Usage (inside gpurun): bash tools/ab/ab_libs.sh [-r REPS] lib1.so lib2.so ...
cd "${GRAFT_REPO_ROOT:-.}"
reps=3
if [ "$1" = "-r" ]; then reps=$2; shift 2; fi
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --steps 200 --warmup 20 --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%7.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in $(seq $reps); do
  for lib in "$@"; do
    printf "%-36s static4096 placed %s plain %s warm %s | trot8192 placed %s | trot65536 placed %s\n" "$lib" \
      "$(run $lib)" "$(run $lib --method plain)" "$(run $lib --method warm)" "$(run $lib --gait trot --batch 8192)" \
      "$(run $lib --gait trot --batch 65536 --steps 50)"
  done
done
