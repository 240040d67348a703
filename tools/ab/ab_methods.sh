#!/bin/bash
# A/B on ONE GPU box: the trajectory-driven bench line by each method (warm / placed / plain) through several builds of the
# library, alternating, so that box-to-box clock differences cancel.
# Usage (inside gpurun): bash tools/ab/ab_methods.sh [-r REPS] [-k STEPS] lib1.so lib2.so ...
cd "${GRAFT_REPO_ROOT:-.}"
reps=2; K=200
while getopts "r:k:" o; do case $o in r) reps=$OPTARG;; k) K=$OPTARG;; esac; done
shift $((OPTIND-1))
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --steps $K --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); t=d['config']['trajectory']
print('%7.2f (k %6.2f%s)' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, ' same %.3f rej %d' % (t['working_set_unchanged'], t['warm_rejected']) if 'working_set_unchanged' in t else ''), end='')"; }
for rep in $(seq $reps); do
  for lib in "$@"; do
    for wl in "--gait static" "--gait trot" "--gait trot --batch 8192" "--gait trot --batch 65536"; do
      printf "%-28s %-28s warm %s | placed %s | plain %s\n" "$(basename $lib)" "$wl" "$(run $lib $wl --method warm)" "$(run $lib $wl --method placed)" "$(run $lib $wl --method plain)"
    done
  done
done
