cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%8.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2; do for lib in "$@"; do printf "%-28s trot warm: 16384 %s | 32768 %s | 65536 %s | static 65536 %s\n" "$(basename $lib)" "$(run $lib --gait trot --batch 16384 --steps 100)" "$(run $lib --gait trot --batch 32768 --steps 100)" "$(run $lib --gait trot --batch 65536 --steps 64)" "$(run $lib --batch 65536 --steps 64)"; done; done
