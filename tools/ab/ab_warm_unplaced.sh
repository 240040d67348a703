cd "${GRAFT_REPO_ROOT:-.}"
for b in ${BATCHES:-4096 8192 16384 32768 65536}; do for g in static trot; do for u in 0 1; do
r=$(QLAMD_BENCH_WARM_UNPLACED=$u python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --gait $g --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), d['config']['all_status_ok'])")
echo "$b $g unplaced=$u $r"; done; done; done
