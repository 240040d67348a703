#!/bin/bash
# A/B on one GPU box: bench the library in the tree ("new") against scratch_bin/libqlamd_base.so ("base"),
# alternating, so that box-to-box clock differences cancel.  Usage (inside gpurun): bash tools/ab/ab_bench.sh
cd "${GRAFT_REPO_ROOT:-.}"
cp quadruped_locomotion_amd/libqlamd.so /tmp/libqlamd_new.so
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%8.2f us/step %8.1f M/s' % (d['ms_per_step']*1e3, d['value']/1e6))"; }
for rep in 1 2; do
  for which in new base; do
    if [ $which = base ]; then cp scratch_bin/libqlamd_base.so quadruped_locomotion_amd/libqlamd.so; else cp /tmp/libqlamd_new.so quadruped_locomotion_amd/libqlamd.so; fi
    echo "$which static4096: $(run)"
    echo "$which trot4096:   $(run --gait trot)"
    echo "$which trot65536:  $(run --gait trot --batch 65536 --steps 50)"
    echo "$which static1M:   $(run --batch 1048576 --steps 30)"
  done
done
cp /tmp/libqlamd_new.so quadruped_locomotion_amd/libqlamd.so
