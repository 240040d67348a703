cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%8.2f' % (d['ms_per_step']*1e3), end='')"; }
for lib in "$@"; do printf "%-28s trot warm: 20480 %s | 24576 %s | 28672 %s\n" "$(basename $lib)" "$(run $lib --gait trot --batch 20480 --steps 100)" "$(run $lib --gait trot --batch 24576 --steps 100)" "$(run $lib --gait trot --batch 28672 --steps 100)"; done
