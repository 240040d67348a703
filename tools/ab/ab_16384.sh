cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%8.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2; do
  for lib in scratch_bin/libqlamd_shadow_prio.so scratch_bin/libqlamd_spill28.so scratch_bin/libqlamd_spill20.so scratch_bin/libqlamd_final.so; do
    printf "%-36s trot 16384 placed %s warm %s | 32768 placed %s warm %s\n" "$lib" "$(run $lib --gait trot --batch 16384)" "$(run $lib --gait trot --batch 16384 --method warm)" "$(run $lib --gait trot --batch 32768)" "$(run $lib --gait trot --batch 32768 --method warm)"
  done
done
