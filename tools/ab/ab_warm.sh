cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); t=d['config']['trajectory']
print('%7.2f (k %6.2f) it %.2f/%d' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['config']['iterations']['mean'], d['config']['iterations']['max']), end='')"; }
for rep in 1 2; do for lib in "$@"; do printf "%-32s static %s | trot %s | trot8192 %s\n" "$(basename $lib)" "$(run $lib)" "$(run $lib --gait trot)" "$(run $lib --gait trot --batch 8192)"; done; done
