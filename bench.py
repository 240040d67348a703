#!/usr/bin/env python3
"""Headline benchmark: control-step QP solves/sec (BASELINE.json `metric`).

One "step" = one balance-controller control tick for every robot of the batch:
virtual-model wrench -> leg FK -> contact-force-distribution QP -> joint torques
(clamped), through the C-ABI (qlamd_balance_solve_batch) with all inputs and
outputs resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--gait static|trot] [--batch B]

N = 1: BASELINE configs[1], batch = 4096 robots, static 4-contact stance.
N > 1: launched by torch.distributed.run, one rank per GPU; every rank solves its
own shard of `batch` robots (weak scaling: robots are independent, no data-path
collective) and the joint torques are all-gathered over RCCL/xGMI for result
collection, as the north star asks.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 408   # 304 B state + 4 B stance in, 96 B torques + 4 B status out
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6
# HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), see
# profiles/r1/hbm_traffic_and_sq_pmc_bench_static_b4096_final.json; None for configurations not profiled.
MEASURED_TRAFFIC_BYTES = {(4096, "static"): int((1108.5 + 416.0) * 1024)}
# VALU wave-instructions per launch from the same file (SQ_INSTS_VALU, its own pass).  A wave64 VALU instruction
# occupies its SIMD16 for 4 cycles, so insts * 4 / (SIMDs * kernel cycles) is the fraction of the chip's VALU
# issue slots the launch used -- the resource this FP64 path is actually bound by.
MEASURED_VALU_INSTS = {(4096, "static"): 1743876}
N_SIMD, SHADER_CLOCK_HZ = 1024, 2.4e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="robots per GPU")
    ap.add_argument("--gait", default="static", choices=["static", "trot"])
    ap.add_argument("--workload", default="balance", choices=["balance", "pose_sqp", "wholebody", "wholebody_dynamics", "full_tick"],
                    help="pose_sqp = BASELINE config 5; wholebody / wholebody_dynamics = SURVEY 8 row f4; full_tick = the "
                         "whole update() from a serialised message to 12 efforts (rows a1 + f1 + f2) "
                         "(all reported separately from the headline metric; single GPU)")
    ap.add_argument("--rpw", type=int, default=0, help="robots per wavefront (0 = auto)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph of K steps")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the multi-GPU code path (process group + all-gather) even with one rank (self-test)")
    ap.add_argument("--overlap-gather", action="store_true",
                    help="capture the gathers on a second stream even with one rank (self-test of the multi-rank graph)")
    ap.add_argument("--no-gather", action="store_true",
                    help="several ranks without the per-step all-gather of the torques (scaling with / without it)")
    ap.add_argument("--ragged", action="store_true", help="full_tick: every message with its own layout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(state, seconds):
    """The oracle (plain-C restatement of the reference path) on the host cores of this box,
    same workload, bounded sample.  Reported next to the GPU number; not the target.
    The thread count is the one that runs fastest here (a container's CPU quota can be far
    below the visible core count); `cores` states it."""
    from oracle import oracle as O
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    B = state["q"].shape[0]

    def rate(threads, budget):
        O.balance_batch(state, nthreads=threads)
        n, t0 = 0, time.perf_counter()
        while True:
            O.balance_batch(state, nthreads=threads)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget or n >= 100000:
                return n * B / dt, n, dt

    cands = sorted({1, 8, 32, visible} | ({visible // 2} if visible >= 4 else set()))
    cands = [c for c in cands if 1 <= c <= visible]
    probe = {c: rate(c, 0.4)[0] for c in cands}
    best = max(probe, key=probe.get)
    value, n, dt = rate(best, seconds)
    return {"value": value, "unit": "control-step QP solves/s", "cores": best, "kind": "port",
            "single_thread_value": probe.get(1), "visible_cores": visible,
            "sample": "%d passes over the same %d-robot batch (%.1f s), OpenMP over robots, %d threads "
                      "(fastest of %s)" % (n, B, dt, best, cands)}


def bench_pose_sqp(args):
    """BASELINE configs[4]: batch pose optimisation, exactly 5 SQP iterations x inner QP per problem."""
    import torch
    from quadruped_locomotion_amd import capi, synth
    B = args.batch
    pb = synth.make_pose_problems(B)
    ctx = capi.Context(device=0)
    prm = capi.default_pose_params()
    prm.tolerance, prm.max_iterations = 0.0, 5
    d = {k: torch.from_numpy(v).to("cuda:0") for k, v in pb.items()}
    out = (torch.zeros(B, 7, dtype=torch.float64, device="cuda:0"), torch.zeros(B, dtype=torch.int32, device="cuda:0"),
           torch.zeros(B, dtype=torch.int32, device="cuda:0"))
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(args.warmup):
        capi.pose_sqp(ctx, d, prm, memory=capi.MEM_DEVICE, out=out, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        capi.pose_sqp(ctx, d, prm, memory=capi.MEM_DEVICE, out=out, stream=stream)
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps
    algo = 424 * B  # 46 doubles in + 7 out per pose-SQP solve
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    cpu = None
    if not args.no_cpu_baseline:
        # the oracle (C restatement, kind "port") on the host cores: same problems, 5 iterations, bounded sample
        from oracle import oracle as O
        visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        _, _, _, probs = O.pose_sqp_batch(pb, synth.POSE_HIPS, synth.POSE_LEG_ORDER, tol=0.0, max_iter=5)

        def rate(threads, budget):
            n, t0 = 0, time.perf_counter()
            while True:
                O.pose_sqp_batch(pb, synth.POSE_HIPS, synth.POSE_LEG_ORDER, tol=0.0, max_iter=5, nthreads=threads, problems=probs)
                n += 1
                dt = time.perf_counter() - t0
                if dt >= budget:
                    return n * B / dt, n, dt
        cands = [c for c in sorted({1, 8, 32, visible} | ({visible // 2} if visible >= 4 else set())) if 1 <= c <= visible]
        probe = {c: rate(c, 0.4)[0] for c in cands}
        best = max(probe, key=probe.get)
        v, n, dt = rate(best, min(args.cpu_seconds, 8.0))
        cpu = {"value": v, "unit": "pose-SQP solves/s", "cores": best, "kind": "port", "single_thread_value": probe.get(1),
               "visible_cores": visible, "sample": "%d passes over the same %d problems (%.1f s), OpenMP over problems, "
               "%d threads (fastest of %s)" % (n, B, dt, best, cands)}
    print(json.dumps({
        "metric": "pose-SQP solves/sec (config 5, reported separately from the headline metric)",
        "value": B * args.steps / elapsed, "unit": "solves/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "batch=%d pose optimisations, 5 SQP iterations x inner Goldfarb-Idnani QP "
                               "(n=6, m=8, dummy equality)" % B, "all_status_ok": bool((out[2] == 0).all().item())},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "pose_sqp_coop_kernel", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": algo},
        **({"cpu_baseline": cpu} if cpu else {})}), flush=True)


def bench_wholebody(args):
    """SURVEY.md section 8 row f4 (no counterpart in the reference): `wholebody` = one whole-body control step per robot
    (inverse dynamics -> force/torque QP -> joint efforts), `wholebody_dynamics` = mass matrix + bias forces + contact
    Jacobian per robot (the HBM-bound kernel: 4464 B written per robot)."""
    import torch
    from quadruped_locomotion_amd import capi, synth
    B, solve = args.batch, args.workload == "wholebody"
    gait = "trot" if args.gait == "trot" else "static"
    s = synth.make_wholebody_states(B, gait)
    ctx = capi.Context(device=0)
    d = capi.to_device(s)
    dev = dict(dtype=torch.float64, device="cuda:0")
    tau, grf, st = torch.zeros(B, 12, **dev), torch.zeros(B, 12, **dev), torch.zeros(B, dtype=torch.int32, device="cuda:0")
    M, h, Jc = torch.zeros(B, 18, 18, **dev), torch.zeros(B, 18, **dev), torch.zeros(B, 12, 18, **dev)
    stream = torch.cuda.current_stream().cuda_stream

    def launch():
        if solve:
            capi.wholebody_solve_device(ctx, d, tau, grf, st, stream=stream)
        else:
            capi.wholebody_dynamics_device(ctx, d, M, h, Jc, stream=stream)

    for _ in range(args.warmup):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps
    # algorithmic bytes per robot: q, qd (2 x 96), quaternion 32, twists 48 in; solve: + a_des 48 + stance 4 in, torques 96 +
    # forces 96 + status 4 out; dynamics: M 2592 + h 144 + Jc 1728 out
    per = (272 + 52 + 196) if solve else (272 + 4464)
    achieved = per * B / (kernel_ms * 1e-3) / 1e9
    cpu = None
    if solve and not args.no_cpu_baseline:
        from oracle import oracle as O
        visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")

        def rate(threads, budget):
            n, t0 = 0, time.perf_counter()
            while True:
                O.wb_step_batch(s, nthreads=threads)
                n += 1
                dt = time.perf_counter() - t0
                if dt >= budget:
                    return n * B / dt, n, dt
        cands = [c for c in sorted({1, 8, 32, visible} | ({visible // 2} if visible >= 4 else set())) if 1 <= c <= visible]
        probe = {c: rate(c, 0.4)[0] for c in cands}
        best = max(probe, key=probe.get)
        v, n, dt = rate(best, min(args.cpu_seconds, 8.0))
        cpu = {"value": v, "unit": "whole-body control steps/s", "cores": best, "kind": "port", "single_thread_value": probe.get(1),
               "visible_cores": visible, "sample": "%d passes over the same %d robots (%.1f s), the oracle's %d-variable QP with "
               "equalities, OpenMP over robots, %d threads (fastest of %s)" % (n, B, dt, 24, best, cands)}
    print(json.dumps({
        "metric": ("whole-body control steps/sec" if solve else "whole-body dynamics evaluations/sec") +
                  " (SURVEY 8 row f4, reported separately from the headline metric)",
        "value": B * args.steps / elapsed, "unit": "steps/s" if solve else "evaluations/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("batch=%d robots, %s, whole-body step: 18-DoF inverse dynamics -> force/torque QP (12 variables "
                                "after eliminating the torques, up to 44 inequality rows) -> 12 joint efforts" if solve else
                                "batch=%d robots, %s, 18x18 mass matrix + bias forces + 12x18 contact Jacobian") % (B, gait),
                   **({"all_status_ok": bool((st == 0).all().item())} if solve else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     # HBM bytes per launch from rocprofv3 --pmc (profiles/r1/pmc_dynamics_b65536.json), measured at 65536 robots
                     "traffic": int((10361.5 + 285696.0) * 1024) if (not solve and B == 65536) else None,
                     "kernel": "wholebody_solve_kernel" if solve else "wholebody_dynamics_kernel",
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": per * B},
        **({"cpu_baseline": cpu} if cpu else {})}), flush=True)


def bench_full_tick(args):
    """The plugin's whole tick for every robot: serialised /desired_robot_state message -> leg state machine -> balance
    solve -> swing branch -> 12 efforts (qlamd_full_tick_batch).  Messages: one per robot; by default copies of one
    message (one publisher's layout: the layout template of the unpack kernel hits from the second launch on, the
    measured states still differ per robot); --ragged gives every message its own layout and payload (every message
    is walked)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from quadruped_locomotion_amd import capi, synth
    from test_wire_format import random_message
    B = args.batch
    rng = np.random.default_rng(11)
    if args.ragged:
        raws = [random_message(rng, ragged=True)[0] for _ in range(B)]
    else:
        base = bytearray(random_message(np.random.default_rng(3), ragged=True)[0])
        raws = [bytes(base)] * B
    off = np.zeros(B + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in raws])
    s = synth.make_states(B, "trot")
    host = dict(messages=np.frombuffer(b"".join(raws), np.uint8).copy(), offsets=off, joint_position=s["q"],
                joint_velocity=rng.normal(scale=0.3, size=(B, 12)), joint_velocity_oldest=rng.normal(scale=0.3, size=(B, 12)),
                base_position=s["base_pos"], base_orientation=s["base_quat"], base_linear_velocity=np.ascontiguousarray(s["base_linvel"]),
                base_angular_velocity=np.ascontiguousarray(s["base_angvel"]), contact=rng.integers(0, 2, (B, 4)).astype(np.uint8),
                limb_state=np.zeros((B, 4), np.int8), store_flag=np.zeros((B, 4), np.uint8), stored_joint_position=np.zeros((B, 12)),
                leg_mode=np.zeros((B, 4), np.uint8), support=np.ones((B, 4), np.uint8), pid_error_last=np.zeros((B, 12)),
                pid_error_integral=np.zeros((B, 12)), joint_effort=np.zeros((B, 12)), leg_state_code=np.zeros((B, 4), np.int8), status=np.full(B, -1, np.int32),
                message_status=np.full(B, -1, np.int32))
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to("cuda:0") for k, v in host.items()}
    ctx = capi.Context(device=0)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(max(args.warmup, 2)):
        capi.full_tick(ctx, dev, 0.0025, memory=capi.MEM_DEVICE, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        capi.full_tick(ctx, dev, 0.0025, memory=capi.MEM_DEVICE, stream=stream)
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tick_ms = e0.elapsed_time(e1) / args.steps
    nbytes = int(off[-1])
    # algorithmic bytes per robot: its message + measured state (q, qd, qd_oldest 288, base pose / twist 104, contact 4) +
    # persistent state read and written (2 x (4 + 4 + 96 + 4 + 4 + 96 + 96)) + efforts 96 + codes / statuses 12
    per = 288 + 104 + 4 + 2 * 304 + 96 + 12
    algo = nbytes + per * B
    achieved = algo / (tick_ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "whole control ticks/sec, message to efforts (SURVEY 8 rows a1 + f1 + f2, reported separately from the headline metric)",
        "value": B * args.steps / elapsed, "unit": "ticks/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "batch=%d robots, one serialised free_gait_msgs/RobotState (%d B on average, %s) per robot and tick, "
                               "trot states: unpack -> leg state machine -> balance solve -> swing branch" %
                               (B, nbytes // B, "ragged layouts" if args.ragged else "one layout"),
                   "messages_ok": int((dev["message_status"] == 0).sum().item()), "solves_ok": int((dev["status"] == 0).sum().item())},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "four kernels per tick (unpack, leg state, balance, swing branch)",
                     "kernel_ms": tick_ms, "algorithmic_bytes_per_launch": algo}}), flush=True)


def main():
    args = parse()
    if args.workload == "full_tick":
        return bench_full_tick(args)
    if args.workload == "pose_sqp":
        return bench_pose_sqp(args)
    if args.workload in ("wholebody", "wholebody_dynamics"):
        return bench_wholebody(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    from quadruped_locomotion_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    B = args.batch
    # rank r owns robots [r*B, (r+1)*B) of the global batch (contiguous shards)
    state = synth.make_states(B, args.gait, offset=rank * B)
    ctx = capi.Context(device=local_rank)
    if args.rpw:
        ctx.set_robots_per_wave(args.rpw)
    d = capi.to_device(state, dev)
    tau = [torch.zeros(B, 12, dtype=torch.float64, device=dev) for _ in range(2)]
    status = torch.full((B,), -1, dtype=torch.int32, device=dev)
    gather = collective and not args.no_gather
    gathered = [torch.zeros(world * B, 12, dtype=torch.float64, device=dev) for _ in range(2)] if gather else None
    stream = torch.cuda.current_stream().cuda_stream

    def step(k, events=None):
        buf = k & 1
        if events is not None:
            events[0].record()
        ctx.balance_solve_device(d, tau[buf], None, status, stream=stream)
        if events is not None:
            events[1].record()
        if gather:
            # result collection only; overlaps with the next step's solve (double-buffered)
            return dist.all_gather_into_tensor(gathered[buf], tau[buf], async_op=True)
        return None

    def fence():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    pending = []
    for k in range(args.warmup):
        w = step(k)
        if w is not None:
            w.wait()
    fence()

    # ---- timed region: exactly K steps -----------------------------------------
    # The K steps (solve, plus the all-gather of the torques when there are several ranks) are
    # captured once into a hipGraph (solves on one stream, gathers on a second one) and replayed: a step is a few
    # tens of microseconds, comparable
    # to one eager launch from Python.  The graph is built outside the timed region; the timed
    # region is one replay = K control steps.  If capture fails the steps are launched eagerly,
    # the all-gather of step k then overlapping the solve of step k+1.
    use_graph = not args.no_graph
    graph = None
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    cap = torch.cuda.current_stream().cuda_stream
                    # The gathers go to a second captured stream: gather k (reads tau[k & 1]) overlaps solve k+1
                    # (writes the other buffer); solve k+2 waits for gather k before it reuses the buffer.
                    # (with a single rank the "gather" is a local copy and the extra graph edges cost more than they
                    # hide -- measured 27.8 vs 22.6 us per step -- so the self-test keeps everything on one stream)
                    overlap = gather and (world > 1 or args.overlap_gather)
                    comm = torch.cuda.Stream() if overlap else None
                    gathered_ev = [None, None]
                    for k in range(args.steps):
                        buf = k & 1
                        if overlap and gathered_ev[buf] is not None:
                            side.wait_event(gathered_ev[buf])
                        ctx.balance_solve_device(d, tau[buf], None, status, stream=cap)
                        if overlap:  # RCCL collectives are capturable; they replay from the graph
                            solved = torch.cuda.Event()
                            solved.record(side)
                            comm.wait_event(solved)
                            with torch.cuda.stream(comm):
                                dist.all_gather_into_tensor(gathered[buf], tau[buf])
                                gathered_ev[buf] = torch.cuda.Event()
                                gathered_ev[buf].record(comm)
                        elif gather:
                            dist.all_gather_into_tensor(gathered[buf], tau[buf])
                    if overlap:
                        side.wait_stream(comm)  # join before the capture ends
            torch.cuda.current_stream().wait_stream(side)
        except Exception as e:  # pragma: no cover - fall back to eager launches
            sys.stderr.write("hipGraph capture failed (%s); eager launches\n" % e)
            graph = None
        if collective:
            # every rank must take the same path, or the collectives would not match up
            okflag = torch.tensor([1 if graph is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(okflag, op=dist.ReduceOp.MIN)
            if int(okflag.item()) == 0:
                graph = None
        if graph is not None:
            graph.replay()  # one untimed replay (instantiation / upload)
            fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev = []
    fence()
    t0 = time.perf_counter()
    if graph is not None:
        e0.record()
        graph.replay()
        e1.record()
    else:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for k in range(args.steps):
            w = step(k, ev[k])
            if w is not None:
                pending.append(w)
                if len(pending) > 1:
                    pending.pop(0).wait()
        for w in pending:
            w.wait()
    fence()
    elapsed = time.perf_counter() - t0

    if collective:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0 and args.steps > 0 and gather:  # the gathered buffer holds every rank's torques in rank order
            last = (args.steps - 1) & 1
            assert torch.equal(gathered[last][:B], tau[last]), "all-gather layout"

    if graph is not None:
        # HIP events around the replay: K kernels back to back, so this average includes the
        # ~1.5 us kernel-to-kernel boundary (an upper bound of the pure kernel duration; the
        # rocprofv3 summary under profiles/ has the exact figure)
        kernel_ms = e0.elapsed_time(e1) / args.steps
    else:
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    st = status.cpu().numpy()
    ok = bool((st == 0).all())

    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        algo_bytes = ALGO_BYTES_PER_STEP * B
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "control-step QP solves/sec (18-DoF, 4-contact) at 1/2/4/8 MI355X",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "batch=%d robots per GPU, %s, one balance-controller control step "
                                   "(virtual-model wrench + leg FK + force-distribution QP + torques) per robot"
                                   % (B, "static 4-contact stance" if args.gait == "static"
                                      else "trot gait (2<->4 contacts)"),
                       "robots_per_gpu": B, "gait": args.gait, "seed": synth.SEED,
                       "result_collection": "rccl all_gather of torques" if gather else
                       ("none (--no-gather)" if collective else "none (single GPU)"),
                       "launch": "hipGraph of K steps" if graph is not None else "eager",
                       "all_status_ok": ok},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": MEASURED_TRAFFIC_BYTES.get((B, args.gait)),
                         "kernel": "balance_coop_kernel", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": algo_bytes},
        }
        insts = MEASURED_VALU_INSTS.get((B, args.gait))
        if insts is not None:
            line["valu_issue"] = {"insts_per_launch": insts, "frac": insts * 4.0 / (N_SIMD * kernel_ms * 1e-3 * SHADER_CLOCK_HZ),
                                  "note": "SQ_INSTS_VALU (rocprofv3 --pmc) x 4 cycles / (1024 SIMDs x kernel cycles at 2.4 GHz)"}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(state, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
