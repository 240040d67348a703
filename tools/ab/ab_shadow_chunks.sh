cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --steps 200 --warmup 20 --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%7.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2 3; do
  for lib in scratch_bin/libqlamd_legs3.so scratch_bin/libqlamd_shadow_prio.so scratch_bin/libqlamd_shadow_2048.so scratch_bin/libqlamd_shadow_4096.so; do
    printf "%-36s static4096 placed %s %s warm %s calm placed %s calm warm %s | trot8192 placed %s warm %s\n" "$lib" \
      "$(run $lib)" "$(run $lib)" "$(run $lib --method warm)" "$(run $lib --errors calm)" "$(run $lib --errors calm --method warm)" "$(run $lib --gait trot --batch 8192)" "$(run $lib --gait trot --batch 8192 --method warm)"
  done
done
