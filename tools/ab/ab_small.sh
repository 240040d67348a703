cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --steps 200 --warmup 20 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%8.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2 3 4 5; do
  for lib in scratch_bin/libqlamd_base.so scratch_bin/libqlamd_spill20.so scratch_bin/libqlamd_cold2048.so; do
    printf "%-36s static4096 placed %s plain %s warm %s | trot8192 placed %s\n" "$lib" "$(run $lib)" "$(run $lib --method plain)" "$(run $lib --method warm)" "$(run $lib --gait trot --batch 8192)"
  done
done
