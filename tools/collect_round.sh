#!/bin/bash
# Everything a round commits under profiles/rN, in one GPU-box call:
#   bash tools/collect_round.sh r6        -> gpurun_out/r6_final/ (copy the summaries into profiles/r6/ afterwards)
# 1. rocprofv3 kernel statistics and PMC passes (tools/collect_profiles.py) -> pmc_index.json, kernel_stats_*.csv
# 2. the bench lines of every workload (the PMC index is put where bench.py looks for it first)
# 3. kernel durations of the auxiliary entries, the single-wavefront latency probe, host-buffer latencies
set -u
R=${1:-r6}
OUT=gpurun_out/${R}_final
mkdir -p $OUT profiles/$R
python3 tools/collect_profiles.py $OUT > $OUT/collect_profiles.log 2>&1
rm -rf $OUT/raw
cp $OUT/pmc_index.json profiles/$R/pmc_index.json
b() { name=$1; shift; python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
b static_b4096 --steps 200 --warmup 20
# the driver's own arguments (K = 20, W = 5): what BENCH_rNN.json will hold
b static_b4096_driver_args --steps 20 --warmup 5
b static_b4096_placed --method placed --steps 200 --warmup 20 --no-cpu-baseline --no-also
b static_b4096_plain --method plain --steps 200 --warmup 20 --no-cpu-baseline --no-also
b static_calm_b4096 --errors calm --steps 200 --warmup 20 --no-cpu-baseline --no-also
b trot_b4096 --gait trot --steps 200 --warmup 20 --no-also
b trot_b8192 --gait trot --batch 8192 --steps 200 --warmup 20 --no-cpu-baseline --no-also
b trot_b65536 --gait trot --batch 65536 --steps 100 --warmup 10 --no-cpu-baseline --no-also
b static_calm_b1048576_plain --gait static --errors calm --batch 1048576 --method plain --steps 20 --warmup 5 --ticks 4 --no-cpu-baseline --no-also
b trot_b8192_force_collective --force-collective --overlap-gather --gait trot --batch 8192 --steps 100 --warmup 10 --no-cpu-baseline
b trot_b8192_force_collective_no_gather --force-collective --no-gather --gait trot --batch 8192 --steps 100 --warmup 10 --no-cpu-baseline
b trot_b8192_force_collective_alternatives --force-collective --overlap-gather --alternatives --gait trot --batch 8192 --steps 100 --warmup 10 --no-cpu-baseline
# where the one-rank collection cost goes -- kernel trace of the solve + in-place all-gather pipeline on two captured streams
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/fc_raw -o fc -- python3 $GRAFT_REPO_ROOT/bench.py --force-collective --overlap-gather --gait trot --batch 8192 --steps 100 --warmup 10 --no-cpu-baseline --replays 3 > /dev/null 2>&1 )
python3 tools/rocpd_kernels.py $(find $OUT/fc_raw -name "*_results.db" | head -1) $OUT/kernel_stats_trot_b8192_force_collective.csv > /dev/null 2>&1
rm -rf $OUT/fc_raw
b pose_sqp_b4096 --workload pose_sqp --steps 200 --warmup 20
b full_tick_b4096 --workload full_tick --steps 100
b full_tick_b16384 --workload full_tick --batch 16384 --steps 50
b full_tick_b65536 --workload full_tick --batch 65536 --steps 40
b wholebody_trot_b4096 --workload wholebody --steps 100
b wholebody_dynamics_b4096 --workload wholebody_dynamics --steps 100
b wholebody_dynamics_b1048576 --workload wholebody_dynamics --batch 1048576 --steps 10
# the kernels of a whole tick: one launch pair at 4096 robots, four launches (parser, state machine, placed balance step, swing branch) at 65 536
for b in 4096 65536; do
  ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/tick_raw -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload full_tick --batch $b --steps 40 --no-cpu-baseline > /dev/null 2>&1 )
  python3 tools/rocpd_kernels.py $(find $OUT/tick_raw -name "*_results.db" | head -1) $OUT/kernel_stats_full_tick_b$b.csv > /dev/null 2>&1
  rm -rf $OUT/tick_raw
done
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/aux_raw -o aux -- python3 $GRAFT_REPO_ROOT/tools/aux_kernels.py > /dev/null 2>&1 )
python3 tools/rocpd_kernels.py $(find $OUT/aux_raw -name "*_results.db" | head -1) $OUT/kernel_stats_aux_entries_b4096.csv > /dev/null 2>&1
rm -rf $OUT/aux_raw
# the same solvers on device-resident inputs (the dense QP, the weighted LSQ entry, the whole-body step: plain, placed, warm)
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/auxd_raw -o auxd -- python3 $GRAFT_REPO_ROOT/tools/experiments/placed_aux_probe.py > /dev/null 2>&1 )
python3 tools/rocpd_kernels.py $(find $OUT/auxd_raw -name "*_results.db" | head -1) $OUT/kernel_stats_aux_device_resident_b4096.csv > /dev/null 2>&1
rm -rf $OUT/auxd_raw
python3 tools/tail_probe.py 2>&1 | grep -v amdgpu > $OUT/single_wavefront_latency.txt
python3 tools/warm_install_probe.py 2>&1 | grep -v amdgpu > $OUT/warm_install_probe_final.txt
python3 tools/latency_b1.py 2>&1 | grep -v amdgpu > $OUT/host_buffer_latency.txt
python3 tools/experiments/trajectory_stats.py 4096 60 2>&1 | grep -v amdgpu > $OUT/trajectory_stats_final.txt
python3 tests/tools/soak_trajectory.py 65536 48 2>&1 | grep -v amdgpu > $OUT/soak_trajectory.txt
# workgroup stamps at the shipped pace (-DQLAMD_BLOCK_STAMPS alone): the phases of the warm-started loop's wavefronts, of the tick's parser blocks
[ -f variants/libqlamd_blockstamps.so ] && ( L=variants/libqlamd_blockstamps.so; python3 tools/stamp_probe_warm_loop.py --phases --lib $L; python3 tools/stamp_probe_warm_loop.py --phases --gait trot --lib $L; python3 tools/stamp_probe_warm_loop.py --phases --gait trot --cold --lib $L; python3 tools/stamp_probe_tick_blocks.py --lib $L; python3 tools/stamp_probe_tick_blocks.py --ragged --lib $L ) 2>&1 | grep -v amdgpu > $OUT/stamps_warm_loop_and_tick.txt
python3 - > $OUT/multi_gpu_cpp_one_rank.txt 2>&1 <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_multi_gpu_cpp as T
from quadruped_locomotion_amd import synth
T.build_demo()
d = tempfile.mkdtemp()
for B, ticks in ((8192, 40), (65536, 16)):   # trajectories: the loop's hints come from earlier ticks
    st = os.path.join(d, "s%d.bin" % B)
    T.write_states(st, synth.trajectory(B, "trot", ticks))
    for every in (1, 8):
        for extra in (("--warm", "--graph"), ("--warm",), (), ("--plain",)):
            p = T.run("--states", st, "--robots", str(B), "--ticks", str(ticks), "--ranks", "1", "--rank", "0", "--steps", "200", "--gather-every", str(every), *extra)
            print(p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr.strip())
PY
ls $OUT
