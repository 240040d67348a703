cd "${GRAFT_REPO_ROOT:-.}"
run() { lib=$1; shift; python tools/experiments/bench_with_lib.py "$lib" --no-cpu-baseline --no-also --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%8.2f' % (d['ms_per_step']*1e3), end='')"; }
for rep in 1 2; do
  for lib in scratch_bin/libqlamd_final.so scratch_bin/libqlamd_thr40000.so; do
    for B in 12288 16384 20480 24576 32768; do
      printf "%-32s trot %6d plain %s placed %s warm %s\n" "$lib" $B "$(run $lib --gait trot --batch $B --method plain)" "$(run $lib --gait trot --batch $B)" "$(run $lib --gait trot --batch $B --method warm)"
    done
  done
done
