# (the define was a three-line experiment in placement_wave's key_of of csrc/balance_kernel.hip -- "if (B < 16384u) key = 0u;" behind
# QLAMD_PLACE_CLASSES_ONLY -- not kept in the source: profiles/r6/ab_warm_policies.txt has what it measured)
cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%7.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2; do for lib in quadruped_locomotion_amd/libqlamd.so variants/libqlamd_classes.so; do printf "%-28s" "$(basename $lib)"; for b in 6144 8192 12288; do printf " | %d static %s trot %s" $b "$(run $lib --batch $b)" "$(run $lib --batch $b --gait trot)"; done; echo; done; done
