cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%7.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2 3; do for lib in variants/libqlamd_prev.so quadruped_locomotion_amd/libqlamd.so; do printf "%-22s trot 65536 %s | static 65536 %s | trot 32768 %s | cold placed trot 65536 %s\n" "$(basename $lib)" "$(run $lib --gait trot --batch 65536)" "$(run $lib --batch 65536)" "$(run $lib --gait trot --batch 32768)" "$(run $lib --gait trot --batch 65536 --method placed)"; done; done
