#!/bin/bash
# the headline line (static 4096, placed loop, K = 200 and the driver's K = 20) through several builds, alternating
cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%7.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2 3 4 5 6; do
  for lib in "$@"; do
    printf "%-36s K=200 %s %s  K=20 %s %s\n" "$lib" "$(run $lib --steps 200 --warmup 20)" "$(run $lib --steps 200 --warmup 20)" "$(run $lib --steps 20 --warmup 5)" "$(run $lib --steps 20 --warmup 5)"
  done
done
