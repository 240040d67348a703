#!/bin/bash
# Everything a round commits under profiles/rN, in one GPU-box call:
#   bash tools/collect_round.sh r5        -> gpurun_out/r5_final/ (copy the summaries into profiles/r5/ afterwards)
# 1. rocprofv3 kernel statistics and PMC passes (tools/collect_profiles.py) -> pmc_index.json, kernel_stats_*.csv
# 2. the bench lines of every workload (the PMC index is put where bench.py looks for it first)
# 3. kernel durations of the auxiliary entries, the single-wavefront latency probe, host-buffer latencies
set -u
R=${1:-r5}
OUT=gpurun_out/${R}_final
mkdir -p $OUT profiles/$R
python3 tools/collect_profiles.py $OUT > $OUT/collect_profiles.log 2>&1
rm -rf $OUT/raw
cp $OUT/pmc_index.json profiles/$R/pmc_index.json
b() { name=$1; shift; python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
b static_b4096 --steps 200 --warmup 20
b static_calm_b4096 --errors calm --steps 200 --warmup 20 --no-cpu-baseline --no-also
b trot_b4096 --gait trot --steps 200 --warmup 20 --no-also
b trot_b8192 --gait trot --batch 8192 --steps 200 --warmup 20 --no-cpu-baseline --no-also
b trot_b65536 --gait trot --batch 65536 --steps 100 --warmup 10 --no-cpu-baseline --no-also
b static_calm_b1048576 --gait static --errors calm --batch 1048576 --steps 20 --warmup 5 --no-cpu-baseline --no-also
b trot_b8192_force_collective --force-collective --overlap-gather --gait trot --batch 8192 --steps 100 --warmup 10 --no-cpu-baseline
b trot_b8192_force_collective_no_gather --force-collective --no-gather --gait trot --batch 8192 --steps 100 --warmup 10 --no-cpu-baseline
b trot_b8192_force_collective_plain --force-collective --overlap-gather --method plain --gait trot --batch 8192 --steps 100 --warmup 10 --no-cpu-baseline --no-alternatives
# the driver's own arguments (K = 20, W = 5): what BENCH_rNN.json will hold
b static_b4096_driver_args --steps 20 --warmup 5
b static_b4096_warm --method warm --steps 200 --warmup 20 --no-cpu-baseline --no-also
# round 5: where the one-rank collection cost goes -- kernel trace of the solve + all-gather pipeline on two captured streams
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/fc_raw -o fc -- python3 $GRAFT_REPO_ROOT/bench.py --force-collective --overlap-gather --gait trot --batch 8192 --steps 100 --warmup 10 --no-cpu-baseline --no-alternatives --replays 3 > /dev/null 2>&1 )
python3 tools/rocpd_kernels.py $(find $OUT/fc_raw -name "*_results.db" | head -1) $OUT/kernel_stats_trot_b8192_force_collective.csv > /dev/null 2>&1
rm -rf $OUT/fc_raw
b pose_sqp_b4096 --workload pose_sqp --steps 200 --warmup 20
b full_tick_b4096 --workload full_tick --steps 100 --method plain
b full_tick_warm_b4096 --workload full_tick --steps 100 --method warm
b full_tick_ragged_b4096 --workload full_tick --ragged --steps 100
b full_tick_b65536 --workload full_tick --batch 65536 --steps 50
b wholebody_static_b4096 --workload wholebody --gait static --steps 100
b wholebody_trot_b4096 --workload wholebody --gait trot --steps 100
b wholebody_dynamics_b4096 --workload wholebody_dynamics --steps 100
b wholebody_dynamics_b65536 --workload wholebody_dynamics --batch 65536 --steps 50
b wholebody_dynamics_b1048576 --workload wholebody_dynamics --batch 1048576 --steps 10
b wholebody_dynamics_row_b1048576 --workload wholebody_dynamics --wholebody-form row --batch 1048576 --steps 10
b full_tick_b16384 --workload full_tick --batch 16384 --steps 50
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/aux_raw -o aux -- python3 $GRAFT_REPO_ROOT/tools/aux_kernels.py > /dev/null 2>&1 )
python3 tools/rocpd_kernels.py $(find $OUT/aux_raw -name "*_results.db" | head -1) $OUT/kernel_stats_aux_entries_b4096.csv > /dev/null 2>&1
rm -rf $OUT/aux_raw
# the same solvers on device-resident inputs (the dense QP, the weighted LSQ entry, the whole-body step: plain, placed, warm)
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/auxd_raw -o auxd -- python3 $GRAFT_REPO_ROOT/tools/experiments/placed_aux_probe.py > /dev/null 2>&1 )
python3 tools/rocpd_kernels.py $(find $OUT/auxd_raw -name "*_results.db" | head -1) $OUT/kernel_stats_aux_device_resident_b4096.csv > /dev/null 2>&1
rm -rf $OUT/auxd_raw
python3 tools/tail_probe.py 2>&1 | grep -v amdgpu > $OUT/single_wavefront_latency.txt
python3 tools/latency_b1.py 2>&1 | grep -v amdgpu > $OUT/host_buffer_latency.txt
python3 tools/write_bw.py 2>&1 | grep -v amdgpu > $OUT/write_bandwidth.txt
./tools/ubench/issue_model > $OUT/issue_model.txt 2>&1
./tools/ubench/rcp_accuracy > $OUT/rcp_accuracy.txt 2>&1
# round 4: issue rate against the rows enabled in EXEC, the row-mix / batch-mix / per-wavefront probes of the balance
# kernel, the segment stamps of the pose kernel (diagnostic build), the C++ multi-GPU host with one rank
./tools/ubench/exec_mask_model > $OUT/exec_mask_model.txt 2>&1
./tools/ubench/tail64_model > $OUT/tail64_model.txt 2>&1
python3 tools/experiments/half_wave_probe.py 2>&1 | grep -v amdgpu > $OUT/half_wave_probe.txt
python3 tools/experiments/row_mix_probe.py 2>&1 | grep -v amdgpu > $OUT/row_mix_probe.txt
python3 tools/experiments/batch_mix_probe.py 2>&1 | grep -v amdgpu > $OUT/batch_mix_probe.txt
python3 tools/experiments/wave_scan.py 2>&1 | grep -v amdgpu > $OUT/wave_scan.txt
python3 tools/experiments/variant_bench.py 2>&1 | grep -v amdgpu > $OUT/variant_bench.txt
[ -f variants/libqlamd_stamps.so ] && python3 tools/stamp_probe_pose.py 2>&1 | grep -v amdgpu > $OUT/pose_sqp_segments.txt
# round 5: placement of robots into wavefronts (the placed entry on the bench batches, the other QP entries through
# qlamd_place_next_call), the head of a launch by argument passing / record layout, the whole tick workgroup by workgroup
python3 tools/experiments/placed_probe.py --grid 2>&1 | grep -v amdgpu > $OUT/placed_probe.txt
python3 tools/experiments/placed_aux_probe.py 2>&1 | grep -v amdgpu > $OUT/placed_aux_probe.txt
python3 tools/experiments/warm_probe.py 2>&1 | grep -v amdgpu > $OUT/warm_probe.txt
python3 tools/experiments/two_leg_probe.py quadruped_locomotion_amd/libqlamd.so 2>&1 | grep -v amdgpu > $OUT/two_leg_probe_final.txt
( ./tools/ubench/launch_head; ./tools/ubench/launch_head_preload ) > $OUT/launch_head.txt 2>&1
[ -f variants/libqlamd_stamps.so ] && ( python3 tools/stamp_probe_tick_blocks.py; python3 tools/stamp_probe_tick_blocks.py --ragged ) 2>&1 | grep -v amdgpu > $OUT/tick_block_stamps.txt
python3 - > $OUT/multi_gpu_cpp_one_rank.txt 2>&1 <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_multi_gpu_cpp as T
from quadruped_locomotion_amd import synth
T.build_demo()
d = tempfile.mkdtemp()
for B in (8192, 65536):
    st = os.path.join(d, "s%d.bin" % B)
    T.write_states(st, synth.make_states(B, "trot"))
    for every in (1, 8):
        for extra in ((), ("--plain",), ("--warm",)):
            p = T.run("--states", st, "--robots", str(B), "--ranks", "1", "--rank", "0", "--steps", "200", "--gather-every", str(every), *extra)
            print(p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr.strip())
PY
ls $OUT
