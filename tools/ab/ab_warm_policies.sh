cd "${GRAFT_REPO_ROOT:-.}"
# the warm-started loop: latency placement (1), throughput placement (2), no placement (batch order)
for b in ${BATCHES:-4096 8192 12288}; do for g in static trot; do
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --gait $g --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), end=' ')"; }
echo "$b $g: latency $(QLAMD_BENCH_POLICY=1 run)| throughput $(QLAMD_BENCH_POLICY=2 run)| none $(QLAMD_BENCH_WARM_UNPLACED=1 run)"
done; done
