#!/bin/bash
# the plain entry (qlamd_balance_solve_batch) through several builds: the headline batch, a million calm robots, trot at 65 536
cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --method plain --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%8.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2; do for lib in "$@"; do printf "%-28s static 4096 %s | K=20 %s | calm 1M %s | trot 65536 %s | trot 4096 %s\n" "$(basename $lib)" "$(run $lib --steps 200)" "$(run $lib --steps 20)" "$(run $lib --errors calm --batch 1048576 --steps 20 --ticks 4)" "$(run $lib --gait trot --batch 65536 --steps 50)" "$(run $lib --gait trot --steps 200)"; done; done
