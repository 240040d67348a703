cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%8.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2; do
  for lib in scratch_bin/libqlamd_base.so scratch_bin/libqlamd_final.so scratch_bin/libqlamd_park.so; do
    printf "%-36s calm1M plain %s | trot65536 plain %s placed %s warm %s | trot16384 placed %s | static4096 placed %s\n" "$lib" \
      "$(run $lib --errors calm --batch 1048576 --steps 20 --warmup 5 --method plain)" "$(run $lib --gait trot --batch 65536 --steps 50 --method plain)" "$(run $lib --gait trot --batch 65536 --steps 50)" "$(run $lib --gait trot --batch 65536 --steps 50 --method warm)" "$(run $lib --gait trot --batch 16384 --steps 100)" "$(run $lib --steps 200 --warmup 20)"
  done
done
