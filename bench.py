#!/usr/bin/env python3
"""Headline benchmark: control-step QP solves/sec (BASELINE.json `metric`).

One "step" = one balance-controller control tick for every robot of the batch:
virtual-model wrench -> leg FK -> contact-force-distribution QP -> joint torques
(clamped), through the C-ABI (qlamd_balance_solve_batch) with all inputs and
outputs resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--gait static|trot] [--batch B]

N = 1: BASELINE configs[1], batch = 4096 robots, static 4-contact stance with SURVEY.md 8(d)'s literal tracking errors
       (0.02 m / 0.05 rad / 0.1); the same line carries an `also` object with every other BASELINE config measured in the
       same process (static-calm and trot at 4096 robots, trot at 8192 and 65 536 robots, the pose SQP at 4096 problems, the
       whole tick and the whole-body step at 4096 robots), each with its kernel time, roofline fraction and PMC provenance,
       `cold_start` / `unplaced` (the same steps with every QP started from the empty working set, placed and through the
       plain entry) and `scale_point` (8192 trot robots per GPU: the workload every line at every N carries).
Every timed region runs on a TRAJECTORY (synth.trajectory): the K captured steps solve K consecutive control ticks of the same
robots (2.5 ms apart: poses integrated with the twists, a trot's gait phase advanced and its support flags recomputed, so
about 1.1 % of the trot robots change their support set every tick), so that every placement and every warm start comes
from the states of EARLIER ticks, never from the states being solved.
N > 1: BASELINE configs[3], 8192 trot robots per GPU (65 536 on 8 GPUs), one rank per GPU; every rank
solves its own contiguous shard (weak scaling: robots are independent, no data-path collective) and the
joint torques are all-gathered over RCCL/xGMI for result collection, as the north star asks.  Started
either by the driver under torch.distributed.run (RANK / WORLD_SIZE in the environment) or from a bare
shell: `python bench.py --gpus N` then starts the N ranks itself as a child torch.distributed.run (the
parent never touches a GPU API and exits with the child's code).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 408   # 304 B state + 4 B stance in, 96 B torques + 4 B status out
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6
# HBM bytes and VALU wave-instructions per launch are NOT measured by this script: they come from rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU each in its own pass; tools/pmc_collect.py) whose summaries are
# committed as profiles/r*/pmc_index.json together with a hash of the kernel sources they were taken on.  A record
# taken on other sources than the ones in this tree is reported as stale and its numbers are dropped.
# A wave64 VALU instruction occupies its SIMD16 for 4 cycles, so insts * 4 / (SIMDs * kernel cycles) is the fraction
# of the chip's VALU issue slots the launch used -- the resource this FP64 path is actually bound by.
N_SIMD, SHADER_CLOCK_HZ = 1024, 2.4e9


def source_hash():
    """sha256 over the kernel sources, the C header and the build recipe (what a PMC record is valid for)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "quadruped_locomotion_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")))
    files += [os.path.join(ROOT, "include", "qlamd.h"), os.path.join(ROOT, "include", "qlamd_robot_constants.h"),
              os.path.join(ROOT, "quadruped_locomotion_amd", "build.py")]      # the code-generation flags per translation unit
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_record(kernel, batch, workload):
    """(record or None, provenance string) from the newest profiles/r*/pmc_index.json."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_index.json")))
    if not cands:
        return None, "no profiles/r*/pmc_index.json"
    idx = json.load(open(cands[-1]))
    rel = os.path.relpath(cands[-1], ROOT)
    for rec in idx.get("records", []):
        if rec["kernel"] == kernel and rec["batch"] == batch and rec["workload"] == workload:
            if rec.get("source_hash") != source_hash():
                return None, "%s: record taken on sources %s, this tree is %s (stale, dropped)" % (
                    rel, rec.get("source_hash"), source_hash())
            return rec, "%s (%s)" % (rel, rec.get("files", ""))
    return None, "%s: no record for %s / %d / %s" % (rel, kernel, batch, workload)


class PeerBuffers:
    """Result collection without a collective: a [2][world][G][B][12] buffer per rank, allocated with hipMalloc (an IPC
    handle needs the base of an allocation, which a tensor of torch's caching allocator is not), its IPC handle
    exchanged once, every peer's buffer mapped; a step's collection is `world` device-to-device copies of this rank's
    shard into slot `rank` of every rank's buffer, on the collection stream.  """

    def __init__(self, group_doubles, rank, world, dev, dist, torch):
        import ctypes as C
        self.rank, self.world, self.dev, self.dist, self.torch = rank, world, dev, dist, torch
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")

        class Handle(C.Structure):
            _fields_ = [("reserved", C.c_char * 64)]
        self.Handle = Handle
        self.hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
        self.hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.hip.hipIpcGetMemHandle.argtypes = [C.POINTER(Handle), C.c_void_p]
        self.hip.hipIpcCloseMemHandle.argtypes = [C.c_void_p]
        self.group_bytes = group_doubles * 8               # one rank's shard of one group of steps
        self.buf_bytes = self.group_bytes * self.world          # one of the two buffers
        # Every step that can fail on one rank only is followed by a consensus: a rank that raised while the others
        # went on into the next collective would hang the whole job.
        self.own = C.c_void_p()
        self.peers = [None] * self.world
        self.failed = False
        h, ok = Handle(), True
        ok = ok and self.hip.hipMalloc(C.byref(self.own), 2 * self.buf_bytes) == 0
        self.peers[self.rank] = self.own.value
        if self.world > 1:
            ok = ok and self.hip.hipIpcGetMemHandle(C.byref(h), self.own) == 0
            handles = [None] * self.world
            self.dist.all_gather_object(handles, (bool(ok), bytes(h)))  # the 64 raw bytes (a c_char field reads as a C string)
            ok = all(flag for flag, _ in handles)
            if ok:
                for r in range(self.world):
                    if r == self.rank:
                        continue
                    hr = Handle()
                    C.memmove(C.byref(hr), handles[r][1], 64)
                    ptr = C.c_void_p()
                    if self.hip.hipIpcOpenMemHandle(C.byref(ptr), hr, 1) != 0:  # 1 = lazy peer access
                        ok = False
                        break
                    self.peers[r] = ptr.value
            agreed = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.dev)
            self.dist.all_reduce(agreed, op=self.dist.ReduceOp.MIN)
            ok = bool(agreed.item())
        if not ok:
            self.close(barrier=False)
            raise RuntimeError("peer buffers: allocation or IPC mapping failed on some rank")

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with HIP error %d" % (what, rc))

    def scatter(self, buf, src_ptr, stream_handle):
        """this rank's shard of buffer `buf` into slot `rank` of every rank's buffer (a failed copy is remembered, not
        raised: the ranks agree on it at the next consensus point, run_preset)"""
        for r in range(self.world):
            dst = self.peers[r] + buf * self.buf_bytes + self.rank * self.group_bytes
            if self.hip.hipMemcpyAsync(dst, src_ptr, self.group_bytes, 4, stream_handle) != 0:  # 4 = hipMemcpyDefault: the runtime takes the devices from the pointers
                self.failed = True

    def agree_no_failure(self):
        flag = self.torch.tensor([1 if self.failed else 0], dtype=self.torch.int32, device=self.dev)
        if self.world > 1:
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
        if int(flag.item()):
            raise RuntimeError("peer buffers: a device-to-device copy failed on some rank")

    def slot_equals(self, buf, slot, tensor):
        import numpy as np
        host = np.empty(self.group_bytes // 8)
        # a failed read-back is a mismatch, not an exception: the caller's ranks agree on the answer in a collective next
        if self.hip.hipMemcpy(host.ctypes.data, self.own.value + buf * self.buf_bytes + slot * self.group_bytes,
                              self.group_bytes, 2) != 0:
            return False
        return bool(np.array_equal(host, tensor.detach().cpu().numpy().ravel()))

    def close(self, barrier=True):
        for r in range(self.world):
            if r != self.rank and self.peers[r]:
                self.hip.hipIpcCloseMemHandle(self.peers[r])
                self.peers[r] = None
        if self.world > 1 and barrier:
            self.dist.barrier()  # nobody frees a buffer a peer still has mapped and in use
        if self.own:
            self.hip.hipFree(self.own)
            self.own = None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="robots per GPU (default 4096 with one GPU, 8192 with several)")
    ap.add_argument("--gait", default=None, choices=["static", "trot"], help="default static with one GPU, trot with several")
    ap.add_argument("--errors", default="survey", choices=["calm", "survey"],
                    help="static stance tracking errors: survey = SURVEY.md 8(d)'s literal 0.02 m / 0.05 rad / 0.1 (default), "
                         "calm = 0.004 / 0.005 / 0.01 (a robot holding its pose: constraints mostly inactive, DESIGN.md 2)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the `also` / `unplaced` / `scale_point` objects (the other presets, methods and configs)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--replays", type=int, default=11,
                    help="timed samples of exactly K steps each (barrier + synchronize on both sides); the median is reported")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed replays of the captured steps for this long before the timed samples: the shader clock of a GPU "
                         "that has just been idle keeps rising for the first ~10 ms of work (samples of K = 20 steps fell from "
                         "0.520 to 0.500 ms over the eleven samples of one run without it)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU-only check of the multi-rank launcher, sharding and gather layout over gloo: no solve, no measurement")
    ap.add_argument("--wholebody-form", default="auto", choices=["auto", "leg", "row"],
                    help="wholebody_dynamics: one lane per leg, 16 lanes per robot, or the library's choice by batch size")
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
    ap.add_argument("--gather-every", type=int, default=1,
                    help="collect the torques of this many consecutive steps with one collective (default 1: one per control step)")
    ap.add_argument("--collect", default="rccl", choices=["rccl", "peer"],
                    help="result collection: rccl = all-gather of the torque shards (default, BASELINE's north star); peer = "
                         "every rank copies its shard into a buffer of every other rank (hipMemcpyAsync into IPC-mapped memory, "
                         "no collective)")
    ap.add_argument("--alternatives-timeout", type=float, default=240.0,
                    help="seconds the `alternatives` runs may take before the line is printed without them")
    ap.add_argument("--alternatives", action="store_true",
                    help="several ranks: also run the same steps with the other ways of collecting the results (one all-gather per 8 "
                         "steps, peer copies) and report them in an `alternatives` object.  Off by default: they are side "
                         "measurements behind a watchdog, and a rank that hangs in one ends with a non-zero exit code")
    ap.add_argument("--no-alternatives", action="store_true", help=argparse.SUPPRESS)  # (the default since round 6)
    ap.add_argument("--no-gather", action="store_true",
                    help="several ranks without the per-step all-gather of the torques (scaling with / without it)")
    ap.add_argument("--ragged", action="store_true", help="full_tick: every message with its own layout")
    ap.add_argument("--method", default="warm", choices=["placed", "plain", "warm"],
                    help="warm (default): the caller's loop of include/qlamd.h -- every step is one qlamd_balance_solve_placed_batch "
                         "call that runs in the placement made during the previous step from the iteration counts of the step before "
                         "it, starts every robot's active-set loop from its final working set of the previous step, reports counts "
                         "and sets and leaves the placement for the next step, all of it inside the timed region; placed: the same "
                         "without the working sets (every QP from the empty set: the headline of round 5, `cold_start` on the line); "
                         "plain: qlamd_balance_solve_batch (robot s in slot s: the headline of rounds 1-4, `unplaced` on the line)")
    ap.add_argument("--ticks", type=int, default=0,
                    help="length T of the trajectory, in control ticks (0 = 200, or 64 beyond 16 384 robots, rounded down to a "
                         "multiple of K).  Step k of the loop solves tick k %% T; a timed region is K steps and consecutive regions "
                         "continue the trajectory, so the jump back to tick 0 -- a discontinuity no 400 Hz loop has, which costs the "
                         "placed / warm-started loop time, never the other way -- falls into one region in T / K")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 4096 if (args.gpus == 1 or args.workload != "balance") else 8192
    if args.gait is None:
        args.gait = "static" if args.gpus == 1 else "trot"
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as a child torch.distributed.run and exit with
    its code.  This parent process must not touch any GPU API (it only counts on the child for that)."""
    import subprocess
    # --standalone: the c10d rendezvous picks a free port itself and hands it to the ranks (nothing bound and released
    # here for another process to take in between)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota_cores():
    """Cores the container's CPU quota serves (cgroup v2 cpu.max, v1 cpu.cfs_quota_us), or None without a quota."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


CPU_PROBE_SECONDS = 1.5


def cpu_baseline_measure(gait, errors, batch, seconds):
    """Runs in the child process started by cpu_baseline() (numpy + the oracle only, threads pinned by the environment
    it was started with).  Every thread count is measured for at least CPU_PROBE_SECONDS -- long enough for a CPU quota
    to throttle an oversubscribed count -- inside ONE OpenMP region (threads keep their block of robots, no fork / join
    between passes), timed in C between two barriers.  `value` is the best SUSTAINED rate: the fastest count of the
    sweep is run again for `seconds`, and if that falls more than 10 % short of its probe the runner-up is given the
    same chance; the larger of the long runs is reported."""
    from oracle import oracle as O
    from quadruped_locomotion_amd import synth
    state = synth.make_states(batch, gait, errors=errors)
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cpu_quota_cores()
    # thread counts beyond twice what the quota serves only measure the throttle
    limit = visible if quota is None else max(1, min(visible, int(2 * quota + 0.5)))
    B = batch
    tile = max(1, -(-64 * limit // B))       # at least 64 robots per thread and pass (same robots, same work per robot)
    big = {k: np_tile(v, tile) for k, v in state.items()}
    Bb = B * tile

    def rate(threads, budget):
        O.balance_batch_repeat(big, 1, threads)
        dt1 = max(O.balance_batch_repeat(big, 1, threads), 1e-6)
        passes = max(1, int(budget / dt1))
        dt = O.balance_batch_repeat(big, passes, threads)
        return passes * Bb / dt, passes, dt

    cands = sorted({1, 2, 4, 8, 16, 32, 64, 128, limit} | ({limit // 2} if limit >= 4 else set()) |
                   ({int(quota)} if quota and quota >= 1 else set()))
    cands = [c for c in cands if 1 <= c <= limit]
    probe = {c: rate(c, CPU_PROBE_SECONDS)[0] for c in cands}
    order = sorted(cands, key=lambda c: -probe[c])
    sustained = {}
    value, best, n, dt = 0.0, order[0], 0, 0.0
    for c in order[:2]:
        v, nn, d = rate(c, seconds if c == order[0] else 0.5 * seconds)
        sustained[str(c)] = v
        if v > value:
            value, best, n, dt = v, c, nn, d
        if value >= 0.9 * max(probe.values()):
            break
    return {"value": value, "unit": "control-step QP solves/s", "cores": best, "kind": "port",
            "single_thread_value": probe.get(1), "per_core_value": value / best, "visible_cores": visible,
            "quota_cores": quota, "cpu_model": cpu_model(),
            "thread_sweep": {str(c): probe[c] for c in cands}, "thread_sweep_seconds_each": CPU_PROBE_SECONDS,
            "sustained": sustained,
            "threads": "OMP_PLACES=%s OMP_PROC_BIND=%s OMP_WAIT_POLICY=%s" % tuple(
                os.environ.get(k, "-") for k in ("OMP_PLACES", "OMP_PROC_BIND", "OMP_WAIT_POLICY")),
            "sample": "%d passes over the same %d-robot batch%s (%.1f s) inside one OpenMP region, robots blocked over "
                      "%d pinned threads; best sustained rate of %s" % (n, B, " tiled x%d" % tile if tile > 1 else "", dt,
                                                                        best, cands)}


def cpu_baseline(gait, errors, batch, seconds):
    """The oracle (plain-C restatement of the reference path, kind "port") on the host cores of this box, same workload,
    bounded sample.  Reported next to the GPU number; not the target.  Measured in a child process of its own, started
    before this process touches the GPU: libgomp reads OMP_PLACES / OMP_PROC_BIND once, when it is loaded, and `import
    torch` loads it -- so the pinning has to be in the environment of a fresh process."""
    import subprocess
    env = dict(os.environ, OMP_PLACES="cores", OMP_PROC_BIND="close", OMP_WAIT_POLICY="passive", OMP_DYNAMIC="false")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--gait", gait, "--errors", errors,
           "--batch", str(batch), "--cpu-seconds", str(seconds)]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, check=True).stdout.decode()
    return json.loads(out.strip().splitlines()[-1])


def np_tile(v, tile):
    import numpy as np
    return np.ascontiguousarray(np.concatenate([v] * tile, axis=0)) if tile > 1 else v


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
    rec, prov = pmc_record("pose_sqp_coop_kernel", B, "pose_sqp")
    print(json.dumps({
        "metric": "pose-SQP solves/sec (config 5, reported separately from the headline metric)",
        "value": B * args.steps / elapsed, "unit": "solves/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "batch=%d pose optimisations, 5 SQP iterations x inner Goldfarb-Idnani QP "
                               "(n=6, m=8, dummy equality)" % B, "all_status_ok": bool((out[2] == 0).all().item())},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": int(rec["fetch_bytes"] + rec["write_bytes"]) if rec and "fetch_bytes" in rec and "write_bytes" in rec else None,
                     "traffic_source": prov, "kernel": "pose_sqp_coop_kernel", "kernel_ms": kernel_ms,
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
    form = {"auto": capi.DYNAMICS_AUTO, "leg": capi.DYNAMICS_LEG, "row": capi.DYNAMICS_ROW}[args.wholebody_form]
    ctx.set_option(capi.OPT_DYNAMICS_FORM, form)
    leg_form = form == capi.DYNAMICS_LEG or (form == capi.DYNAMICS_AUTO and B > 8192)
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
    kname = "wholebody_solve_kernel" if solve else ("wholebody_dynamics_leg_kernel" if leg_form else "wholebody_dynamics_kernel")
    rec, traffic_src = pmc_record(kname, B, ("wholebody-" + gait) if solve else "wholebody_dynamics")
    traffic = int(rec["fetch_bytes"] + rec["write_bytes"]) if rec and "fetch_bytes" in rec and "write_bytes" in rec else None
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
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": kname,
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": per * B},
        **({"cpu_baseline": cpu} if cpu else {})}), flush=True)


def selftest_launcher(args):
    """CPU-only check of what `--gpus N` does around the solve: ranks started by the same launcher, contiguous shards
    of the seeded generator, the all-gather layout and the max-over-ranks reduction, over gloo.  There is no solve
    and no measurement here (the product path has no CPU fallback): the gathered quantity is the joint-position
    shard itself.  Rank 0 prints one JSON line with "selftest": "launcher"."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from quadruped_locomotion_amd import synth
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = args.batch
    shard = torch.from_numpy(synth.make_states(B, args.gait, offset=rank * B, errors=args.errors)["q"])
    gathered = torch.zeros(world * B, 12, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, shard)
    ones = torch.ones(1, dtype=torch.int32)
    dist.all_reduce(ones)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        whole = synth.make_states(world * B, args.gait, errors=args.errors)["q"]
        print(json.dumps({"selftest": "launcher", "n_gpus": args.gpus, "world_size": world, "ranks_seen": int(ones.item()),
                          "max_over_ranks_ok": t.item() == float(world), "robots_per_rank": B, "gait": args.gait,
                          "gather_layout_ok": bool(np.array_equal(gathered.numpy(), whole)), "value": None}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_measure(args.gait, args.errors, args.batch, args.cpu_seconds)), flush=True)
        return None
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare shell: this process stays off the GPU and starts the ranks as a child
        sys.exit(launch_ranks(args))
    if args.selftest_launcher:
        return selftest_launcher(args)
    if args.workload == "pose_sqp":
        return bench_pose_sqp(args)
    if args.workload == "wholebody_dynamics":
        return bench_wholebody(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # rank 0 at N = 1 only: the CPU baseline, in a pinned child process, BEFORE this process touches the GPU
    cpu = None
    if not args.no_cpu_baseline and world == 1 and args.workload == "balance":
        try:
            cpu = cpu_baseline(args.gait, args.errors, args.batch, args.cpu_seconds)
        except Exception as e:  # the GPU measurement must not be lost to a failing host-side probe
            sys.stderr.write("cpu_baseline failed (%s: %s); the line is printed without it\n" % (type(e).__name__, e))

    import numpy as np
    import torch
    import torch.distributed as dist

    from quadruped_locomotion_amd import capi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    collective = world > 1 or args.force_collective
    ranks_seen = 1
    json_fd = 1
    if collective:
        # RCCL writes a version banner to file descriptor 1 when it shuts down; the contract is ONE JSON line on stdout.
        # Everything this process and its libraries write to fd 1 from here on goes to stderr; the line itself is
        # written to the saved descriptor.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)  # every rank answers over RCCL
        ranks_seen = int(ones.item())
        assert ranks_seen == world == dist.get_world_size(), "RCCL saw %d ranks, expected %d" % (ranks_seen, world)

    B = args.batch
    ctx = capi.Context(device=local_rank)
    if args.rpw:
        ctx.set_robots_per_wave(args.rpw)
    gather = collective and not args.no_gather
    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        torch.cuda.synchronize()
        if collective:  # the barrier is device work of its own (an all-reduce): synchronize again behind it
            dist.barrier()
            torch.cuda.synchronize()

    trajectories = {}   # (gait, errors, batch, ticks, rank) -> (device-resident states of every tick, support switches per tick)

    def run_preset(gait, errors, with_gather, replays, second_without_gather, collect="rccl", every=1, batch=None,
                   method="plain", records=False):
        """One workload preset on this rank's shard: warm-up, K steps captured as one hipGraph, `replays` timed samples.
        every: steps per collection; collect: "rccl" (all-gather) or "peer" (copies into the peers' buffers).
        Step k of the loop solves tick k % T of a T-tick trajectory of the shard's robots (synth.trajectory).  method "placed": the caller's loop of include/qlamd.h -- step k runs in order[k & 1], writes
        iters[k & 1] and makes order[(k + 1) & 1] from iters[(k - 1) & 1]: tick k runs in the placement made during tick
        k - 1 from the counts of tick k - 2; "warm": the same, and every robot's QP starts from its final working set of tick
        k - 1 (one array, updated in place)."""
        B = batch or args.batch
        K = args.steps
        # The trajectory: T consecutive ticks, 200 by default (a tenth of that beyond 16 384 robots: 20 MB a tick at 65 536), a
        # multiple of K when it is longer than a timed region.  A timed region is exactly K steps; consecutive regions -- and the
        # untimed replays between them -- CONTINUE the trajectory (NG = T / K captured graphs of K steps each, replayed in turn),
        # so the jump back to tick 0, which no 400 Hz loop has, falls into one region in NG, not into every one.
        cap = 200 if B <= 16384 else 64
        T = max(1, args.ticks) if args.ticks else (cap if K >= cap else K * max(1, cap // K))
        NG = max(1, -(-T // K)) if K > 0 else 1
        # rank r owns robots [r*B, (r+1)*B) of the global batch (contiguous shards)
        # records: the double fields of a robot in one 48-double record (QLAMD_OPT_STATE_LAYOUT = QLAMD_STATE_RECORDS) instead of
        # one array per field -- what a permuted launch gathers in three lines instead of nine
        key = (gait, errors, B, T, rank, records)
        if key not in trajectories:
            states = synth.trajectory(B, gait, T, offset=rank * B, errors=errors)
            put = capi.to_device_records if records else capi.to_device
            trajectories[key] = ([put(st_, dev) for st_ in states], synth.support_switches(states + [states[0]]))
        ds, switched = trajectories[key]   # (switched: T - 1 transitions of the trajectory, then the jump back to tick 0)
        ctx.set_option(capi.OPT_STATE_LAYOUT, capi.STATE_RECORDS if records else capi.STATE_FIELDS)
        retries0, giveups0 = ctx.counter(capi.COUNTER_WARM_RETRIES), ctx.counter(capi.COUNTER_PLACEMENT_GIVE_UPS)
        placed, warm = method == "placed", method == "warm"
        orders = [torch.arange(B, dtype=torch.int32, device=dev) for _ in range(2)] if (placed or warm) else None
        its = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)] if (placed or warm) else None
        wset = torch.zeros(B, dtype=torch.int32, device=dev) if warm else None   # working sets: one array, in place

        def solve(k, out, st):
            if warm and os.environ.get("QLAMD_BENCH_WARM_UNPLACED") == "1":   # (experiment: the warm start in batch order)
                ctx.balance_solve_placed_device(ds[k % T], out, None, status, iterations=its[k & 1], prev_working_set=wset, working_set=wset, stream=st)
            elif warm:
                # the placed loop, and every robot's active-set loop starts from its final working set of the step before
                # (include/qlamd.h: both halves of the hint a 400 Hz caller has)
                ctx.balance_solve_placed_device(ds[k % T], out, None, status, order=orders[k & 1], iterations=its[k & 1],
                                                prev_iterations=its[(k - 1) & 1], next_order=orders[(k + 1) & 1],
                                                policy=int(os.environ.get("QLAMD_BENCH_POLICY", capi.PLACEMENT_AUTO)),
                                                prev_working_set=wset, working_set=wset, stream=st)
            elif placed:
                ctx.balance_solve_placed_device(ds[k % T], out, None, status, order=orders[k & 1], iterations=its[k & 1],
                                                prev_iterations=its[(k - 1) & 1], next_order=orders[(k + 1) & 1],
                                                policy=capi.PLACEMENT_AUTO, stream=st)
            else:
                ctx.balance_solve_device(ds[k % T], out, None, status, stream=st)
        G = max(1, every)
        status = torch.full((B,), -1, dtype=torch.int32, device=dev)
        peer = PeerBuffers(G * B * 12, rank, world, dev, dist, torch) if (with_gather and collect == "peer") else None
        gathered = ([torch.zeros(world, G, B, 12, dtype=torch.float64, device=dev) for _ in range(2)]
                    if (with_gather and peer is None) else None)
        # Result collection IN PLACE: a step writes its efforts straight into this rank's block of the gathered buffer and the
        # all-gather gets that block as its input (sendbuff = recvbuff + rank x count: RCCL's in-place form, which moves nothing
        # locally).  Out of place, one rank's "gather" was a copy kernel plus two event edges per step (+6.9 us per step, round 5).
        tau = ([gathered[b_][rank] for b_ in range(2)] if gathered is not None else
               [torch.zeros(G, B, 12, dtype=torch.float64, device=dev) for _ in range(2)])

        def collect_now(k):  # the last step of its group, or the last step of all
            return (k % G == G - 1) or (k == args.steps - 1)

        def do_collect(buf, stream_handle):
            if peer is not None:
                peer.scatter(buf, tau[buf].data_ptr(), stream_handle)
                return None
            return dist.all_gather_into_tensor(gathered[buf].view(world * G * B, 12), tau[buf].view(G * B, 12), async_op=True)

        eager_k = [0]   # steps launched eagerly so far: the loop goes on where it is (ticks eager_k % T)

        def step(k, wg, events=None):
            buf = (k // G) & 1
            if events is not None:
                events[0].record()
            solve(eager_k[0], tau[buf][k % G], stream)
            eager_k[0] += 1
            if events is not None:
                events[1].record()
            if wg and collect_now(k):
                # result collection only; overlaps with the next step's solve (double-buffered)
                return do_collect(buf, stream)
            return None

        for k in range(args.warmup):
            w = step(k, with_gather)
            if w is not None:
                w.wait()
        fence()
        if peer is not None:
            peer.agree_no_failure()

        # ---- K steps as one hipGraph ------------------------------------------------------------
        # The K steps (solve, plus the all-gather of the torques when there are several ranks) are captured once into a
        # hipGraph (solves on one stream, gathers on a second one) and replayed: a step is a few tens of microseconds,
        # comparable to one eager launch from Python.  If capture fails the steps are launched eagerly, the all-gather
        # of step k then overlapping the solve of step k+1.
        def capture(wg, g):
            """steps g K .. g K + K - 1 of the loop (ticks (g K + j) % T) as one graph"""
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                # (thread_local: the process group's watchdog thread polls its events with hipEventQuery while this thread
                # captures; under the default "global" mode that query invalidates the capture -- seen once in five rounds,
                # profiles/r5/bench_trot_b8192_force_collective.err of the first r5 collection)
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local" if collective else "global"):
                    cap = torch.cuda.current_stream().cuda_stream
                    # The gathers go to a second captured stream: gather k (reads tau[k & 1]) overlaps solve k+1 (writes
                    # the other buffer); solve k+2 waits for gather k before it reuses the buffer.  (With a single rank
                    # the "gather" is a local copy and the extra graph edges cost more than they hide -- measured 27.8
                    # vs 22.6 us per step -- so the self-test keeps everything on one stream unless asked.)
                    overlap = wg and (world > 1 or args.overlap_gather)
                    comm = torch.cuda.Stream() if overlap else None
                    gathered_ev = [None, None]
                    for k in range(args.steps):
                        buf = (k // G) & 1
                        if overlap and k % G == 0 and gathered_ev[buf] is not None:
                            side.wait_event(gathered_ev[buf])
                        solve(g * K + k, tau[buf][k % G], cap)
                        if not (wg and collect_now(k)):
                            continue
                        if overlap:  # RCCL collectives and peer copies are capturable; they replay from the graph
                            solved = torch.cuda.Event()
                            solved.record(side)
                            comm.wait_event(solved)
                            with torch.cuda.stream(comm):
                                if peer is not None:
                                    peer.scatter(buf, tau[buf].data_ptr(), comm.cuda_stream)
                                else:
                                    dist.all_gather_into_tensor(gathered[buf].view(world * G * B, 12), tau[buf].view(G * B, 12))
                                gathered_ev[buf] = torch.cuda.Event()
                                gathered_ev[buf].record(comm)
                        elif peer is not None:
                            peer.scatter(buf, tau[buf].data_ptr(), cap)
                        else:
                            dist.all_gather_into_tensor(gathered[buf].view(world * G * B, 12), tau[buf].view(G * B, 12))
                    if overlap:
                        side.wait_stream(comm)  # join before the capture ends
            torch.cuda.current_stream().wait_stream(side)
            return graph, overlap

        def build_graphs(wg):
            graphs, overlap = None, False
            if not args.no_graph:
                try:
                    graphs = []
                    for g in range(NG):
                        gr, overlap = capture(wg, g)
                        graphs.append(gr)
                except Exception as e:  # pragma: no cover - fall back to eager launches
                    sys.stderr.write("hipGraph capture failed (%s); eager launches\n" % e)
                    graphs = None
                if collective:
                    # every rank must take the same path, or the collectives would not match up
                    okflag = torch.tensor([1 if graphs is not None else 0], dtype=torch.int32, device=dev)
                    dist.all_reduce(okflag, op=dist.ReduceOp.MIN)
                    if int(okflag.item()) == 0:
                        graphs = None
            return graphs, overlap

        def measure(wg):
            graphs, overlap = build_graphs(wg)
            turn = [0]

            def replay():   # the next K steps of the loop
                graphs[turn[0] % NG].replay()
                turn[0] += 1

            if graphs is not None:
                for _ in range(NG):
                    replay()  # one untimed replay of each (instantiation / upload)
                fence()
                # ... and untimed replays until the clocks have settled (every rank the same number: no host clock involved
                # in the count once it is agreed on)
                t_one = time.perf_counter()
                replay()
                fence()
                t_one = max(time.perf_counter() - t_one, 1e-5)
                n_settle = int(min(2000, max(0.0, args.settle_ms * 1e-3) / t_one))
                n_settle += (-(turn[0] + n_settle)) % NG      # the first timed region starts at tick 0
                if collective:
                    nt = torch.tensor([n_settle], dtype=torch.int32, device=dev)
                    dist.all_reduce(nt, op=dist.ReduceOp.MAX)
                    n_settle = int(nt.item())
                for _ in range(n_settle):
                    replay()
                fence()

            def sample():
                """One timed region: exactly K steps between barrier + synchronize on both sides.  Returns (wall seconds,
                mean solve-kernel ms from HIP events on the launch stream)."""
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev, pending = [], []
                fence()
                t0 = time.perf_counter()
                if graphs is not None:
                    replay()  # nothing else inside the timed region: the event pair is taken on a replay of its own below
                else:
                    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                    for k in range(args.steps):
                        w = step(k, wg, ev[k])
                        if w is not None:
                            pending.append(w)
                            if len(pending) > 1:
                                pending.pop(0).wait()
                    for w in pending:
                        w.wait()
                fence()
                elapsed = time.perf_counter() - t0
                if graphs is not None:
                    # HIP events around an untimed replay (the NEXT K steps of the loop): K kernels back to back, so this average
                    # includes the kernel-to-kernel boundary (an upper bound of the pure kernel duration; the rocprofv3 summary
                    # under profiles/ has the exact figure)
                    e0.record()
                    replay()
                    e1.record()
                    fence()
                    kernel_ms = e0.elapsed_time(e1) / args.steps
                else:
                    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
                return elapsed, kernel_ms

            n = max(1, replays)
            el = np.zeros(n)
            km = np.zeros(n)
            for r in range(n):
                el[r], km[r] = sample()
                # A sample takes two replays (the timed one, the one between HIP events): with an even number NG of regions the
                # timed samples would visit every other region of the trajectory only -- and the one behind the jump back to tick 0
                # three times in eleven.  One more untimed replay makes the stride 3: every region in turn (all ranks alike).
                if graphs is not None and NG > 1 and NG % 2 == 0 and NG % 3 != 0:
                    replay()
                    fence()
            if collective:  # a sample lasts as long as its slowest rank
                t = torch.from_numpy(el).to(dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.cpu().numpy()
            pick = int(np.argsort(el)[n // 2])  # the median sample is the timed region reported
            return dict(elapsed=float(el[pick]), kernel_ms=float(np.median(km)), samples_ms=[float(x * 1e3) for x in el],
                        graph=graphs is not None, overlap=overlap)

        res = measure(with_gather)
        if peer is not None:
            peer.agree_no_failure()
        if collective and args.steps > 0 and with_gather:
            # the collected buffer holds every rank's torques in rank order: every rank sends a checksum of the block it computed
            # and checks every block it received against its sender's (peer copies: its own slot against its own buffer)
            last = ((args.steps - 1) // G) & 1
            torch.cuda.synchronize()
            if peer is not None:
                mine = peer.slot_equals(last, rank, tau[last])
            else:
                sums = torch.zeros(world, dtype=torch.float64, device=dev)
                dist.all_gather_into_tensor(sums, tau[last].abs().sum().reshape(1))
                got = gathered[last].abs().sum(dim=(1, 2, 3))
                mine = bool(torch.equal(got, sums)) and bool((sums > 0).all().item())
            flag = torch.tensor([1 if mine else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            assert int(flag.item()) == 1, "collected layout"
            res["gather_layout_ok"] = True
        # (after the check: these steps write over this rank's block of the gathered buffers)
        res["plain"] = measure(False) if (with_gather and second_without_gather) else None
        if peer is not None:
            peer.close()
        # ---- what the loop saw, tick by tick (untimed): the K steps once more, eagerly, continuing the caller's loop where the
        # last replay left it -- every tick's statuses, iteration counts and (warm) how many robots ended the tick with the
        # working set they started it with
        ok, it_mean, it_max, unchanged = True, [], 0, []
        for k in range(T):
            before = wset.clone() if warm else None
            solve(k, tau[0][0], stream)
            torch.cuda.synchronize()
            ok = ok and bool((status == 0).all().item())
            if placed or warm:
                it = its[k & 1]
                it_mean.append(float(it.float().mean().item()))
                it_max = max(it_max, int(it.max().item()))
            if warm:
                unchanged.append(float((wset == before).float().mean().item()))
        if collective:
            okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = bool(okt.item())
        res["ok"] = ok
        res["batch"], res["method"] = B, method
        wrap, inside = [0], list(range(1, T)) or [0]                  # tick 0 follows the jump back from tick T - 1
        res["trajectory"] = {
            "states_per_replay": K, "ticks": T, "regions": NG, "dt": synth.CONTROL_PERIOD,
            "switched_per_tick": float(np.mean(switched[:-1])) if T > 1 else 0.0,   # robots whose support set changes, per tick
            "switched_at_wrap": float(switched[-1])}
        if placed or warm:
            res["iterations"] = {"mean": float(np.mean(it_mean)), "max": it_max}
            res["trajectory"]["placement_give_ups"] = ctx.counter(capi.COUNTER_PLACEMENT_GIVE_UPS) - giveups0
        if warm:
            res["trajectory"]["working_set_unchanged"] = float(np.mean([unchanged[k] for k in inside]))
            res["trajectory"]["working_set_unchanged_at_wrap"] = float(np.mean([unchanged[k] for k in wrap]))
            # robots whose warm start was rejected and that were solved again cold inside the same launch (whole preset:
            # warm-up, settling and every timed replay included); with the fallback on none of them is ever reported
            res["trajectory"]["warm_rejected"] = ctx.counter(capi.COUNTER_WARM_RETRIES) - retries0
        ctx.set_option(capi.OPT_STATE_LAYOUT, capi.STATE_FIELDS)
        return res

    def roofline_of(res, gait, errors):
        """roofline and valu_issue objects of one preset from its kernel time and the committed PMC record."""
        kernel_ms, Bp, placed, warm = res["kernel_ms"], res["batch"], res["method"] == "placed", res["method"] == "warm"
        # placed: + robot_order in, iterations out, prev_iterations in, next_robot_order out (4 B each);
        # warm: the placed loop's four + previous working set in, working set out
        algo_bytes = (ALGO_BYTES_PER_STEP + (16 if placed else 24 if warm else 0)) * Bp
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        wl = ("static-%s" % errors if gait == "static" else "trot") + ("+placed" if placed else "+warm" if warm else "")
        rec, prov = pmc_record("balance_coop_kernel", Bp, wl)
        have = rec is not None and "fetch_bytes" in rec and "write_bytes" in rec
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": int(rec["fetch_bytes"] + rec["write_bytes"]) if have else None, "traffic_source": prov,
                "traffic_rule": rec.get("fetch_size_rule") if have else None,
                "kernel": "balance_coop_kernel", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes}
        valu = None
        if rec and rec.get("valu_insts"):
            insts = rec["valu_insts"]
            valu = {"insts_per_launch": insts, "frac": insts * 4.0 / (N_SIMD * kernel_ms * 1e-3 * SHADER_CLOCK_HZ), "source": prov,
                    "note": "SQ_INSTS_VALU (rocprofv3 --pmc) x 4 cycles / (1024 SIMDs x kernel cycles at 2.4 GHz)"}
        return roof, valu

    def entry(r, gait, errors, note=None):
        """One `also` / `scale_point` record of a finished preset."""
        roof, valu = roofline_of(r, gait, errors)
        e = {"value": r["batch"] * args.steps / r["elapsed"], "unit": "solves/s", "ms_per_step": r["elapsed"] / args.steps * 1e3,
             "kernel_ms": r["kernel_ms"], "batch": r["batch"], "method": r["method"],
             "tracking_error": list(synth.tracking_error(gait, errors)),
             "all_status_ok": r["ok"], "roofline_frac": roof["frac"], "traffic": roof["traffic"],
             "valu_issue_frac": valu["frac"] if valu else None, "pmc_source": roof["traffic_source"]}
        if "iterations" in r:
            e["iterations"] = r["iterations"]
        e["trajectory"] = r["trajectory"]
        if note:
            e["note"] = note
        return e

    def time_captured(step_fn, eager_steps, regions=1):
        """`eager_steps` untimed eager steps, then the loop's steps captured as `regions` hipGraphs of K steps each (step_fn(k, stream)
        with k running on from graph to graph) and replayed in turn: 50 ms of untimed replays, five timed samples of exactly K steps
        between synchronisations (median), the kernel time from HIP events around one more replay."""
        for k in range(eager_steps):
            step_fn(k, stream)
        torch.cuda.synchronize()
        if args.no_graph:   # eager launches (the counter passes of tools/collect_profiles.py): K steps between synchronisations
            el, e0, e1 = [], torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for r in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e0.record()
                for k in range(args.steps):
                    step_fn(r * args.steps + k, stream)
                e1.record()
                torch.cuda.synchronize()
                el.append(time.perf_counter() - t0)
            return float(np.median(el)), e0.elapsed_time(e1) / args.steps
        graphs = []
        for g in range(regions):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    cap = torch.cuda.current_stream().cuda_stream
                    for k in range(args.steps):
                        step_fn(g * args.steps + k, cap)
            torch.cuda.current_stream().wait_stream(side)
            graphs.append(graph)
        turn = [0]

        def replay():
            graphs[turn[0] % regions].replay()
            turn[0] += 1
        for _ in range(regions):
            replay()
        torch.cuda.synchronize()
        t_end = time.perf_counter() + 0.05
        while time.perf_counter() < t_end or turn[0] % regions:
            replay()
        torch.cuda.synchronize()
        el = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            replay()
            torch.cuda.synchronize()
            el.append(time.perf_counter() - t0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        replay()
        e1.record()
        torch.cuda.synchronize()
        return float(np.median(el)), e0.elapsed_time(e1) / args.steps

    def ticks_for(cap):
        """(T, regions) of a side measurement: about `cap` ticks, a multiple of K"""
        K = args.steps
        if args.ticks:
            return max(1, args.ticks), max(1, -(-args.ticks // K))
        return (cap, 1) if K >= cap else (K * max(1, cap // K), max(1, cap // K))

    def pose_sqp_entry(batch):
        """BASELINE configs[4] inside the headline line: batch pose optimisations, exactly 5 SQP iterations each, the K calls
        captured and timed like every other preset."""
        pb = synth.make_pose_problems(batch)
        prm = capi.default_pose_params()
        prm.tolerance, prm.max_iterations = 0.0, 5
        dp = {k: torch.from_numpy(v).to(dev) for k, v in pb.items()}
        out = (torch.zeros(batch, 7, dtype=torch.float64, device=dev), torch.zeros(batch, dtype=torch.int32, device=dev),
               torch.zeros(batch, dtype=torch.int32, device=dev))
        elapsed, kernel_ms = time_captured(lambda k, st_: capi.pose_sqp(ctx, dp, prm, memory=capi.MEM_DEVICE, out=out, stream=st_),
                                           max(2, min(args.warmup, 5)))
        rec, prov = pmc_record("pose_sqp_coop_kernel", batch, "pose_sqp")
        frac = 424.0 * batch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS  # 46 doubles in + 7 out per solve
        return {"value": batch * args.steps / elapsed, "unit": "pose-SQP solves/s", "ms_per_step": elapsed / args.steps * 1e3,
                "kernel_ms": kernel_ms, "batch": batch, "method": "qlamd_pose_sqp_batch, 5 SQP iterations x inner QP (n = 6, m = 8, dummy equality)",
                "all_status_ok": bool((out[2] == 0).all().item()), "roofline_frac": frac,
                "traffic": int(rec["fetch_bytes"] + rec["write_bytes"]) if rec and "fetch_bytes" in rec and "write_bytes" in rec else None,
                "valu_issue_frac": (rec["valu_insts"] * 4.0 / (N_SIMD * kernel_ms * 1e-3 * SHADER_CLOCK_HZ)) if rec and rec.get("valu_insts") else None,
                "pmc_source": prov}

    def full_tick_entry(batch):
        """The plugin's whole tick (rows a1 + f1 + f2: ros_balance_controller.cpp:198-718) on a trot trajectory: per robot and tick
        one serialised /desired_robot_state message (one publisher's layout; desired base state and support flags of the
        trajectory's tick) -> leg state machine -> balance solve -> swing branch -> 12 efforts, qlamd_full_tick_batch, the
        controller's state carried from tick to tick; cold, and with the working set the tick keeps (`warm`)."""
        T, regions = ticks_for(60 if batch <= 16384 else 8)   # (a tick of 65 536 messages is 207 MB)
        states = synth.trajectory(batch, "trot", T)
        rng = np.random.default_rng(11)
        from quadruped_locomotion_amd import wire
        mt = synth.MessageTemplate([wire.MODE_NAMES[k] for k in rng.integers(0, len(wire.MODE_NAMES), 4)])
        fixed = {k: rng.normal(size=(batch, n)) for k, n in mt.DOUBLES}
        fixed["phase"] = rng.random((batch, 4))
        shared_in = dict(joint_position=states[0]["q"], joint_velocity=rng.normal(scale=0.3, size=(batch, 12)),
                         joint_velocity_oldest=rng.normal(scale=0.3, size=(batch, 12)),
                         base_linear_velocity=np.ascontiguousarray(states[0]["base_linvel"]),
                         base_angular_velocity=np.ascontiguousarray(states[0]["base_angvel"]),
                         contact=rng.integers(0, 2, (batch, 4)).astype(np.uint8))
        shared_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in shared_in.items()}
        per_tick, nbytes = [], 0
        for s_ in states:
            f = dict(fixed, des_pos=s_["des_pos"], des_quat=s_["des_quat"], des_linvel=s_["des_linvel"], des_angvel=s_["des_angvel"],
                     support_leg=s_["stance"])
            blob, off = mt.pack(f)
            nbytes = int(off[-1])
            per_tick.append(dict(shared_in, messages=torch.from_numpy(blob).to(dev), offsets=torch.from_numpy(off).to(dev),
                                 base_position=torch.from_numpy(s_["base_pos"]).to(dev),
                                 base_orientation=torch.from_numpy(s_["base_quat"]).to(dev)))
        out = {}
        for warm in (False, True):
            keep = dict(limb_state=np.zeros((batch, 4), np.int8), store_flag=np.zeros((batch, 4), np.uint8),
                        stored_joint_position=np.zeros((batch, 12)), leg_mode=np.zeros((batch, 4), np.uint8),
                        support=np.ones((batch, 4), np.uint8), pid_error_last=np.zeros((batch, 12)), pid_error_integral=np.zeros((batch, 12)),
                        joint_effort=np.zeros((batch, 12)), leg_state_code=np.zeros((batch, 4), np.int8), status=np.full(batch, -1, np.int32),
                        message_status=np.full(batch, -1, np.int32), command=np.zeros(capi.tick_command_bytes(batch), np.uint8))
            if warm:
                keep["working_set"] = np.zeros(batch, np.uint32)
            if batch > 16384:   # the placed loop on the tick's own state (four launches per tick from here on)
                keep["placement_state"] = np.zeros((4, batch), np.int32)
            keep = {k: torch.from_numpy(v).to(dev) for k, v in keep.items()}
            ios = [dict(pt, **keep) for pt in per_tick]
            tctx = capi.Context(device=local_rank)
            tctx.reserve(batch)
            elapsed, tick_ms = time_captured(lambda k, st_: capi.full_tick(tctx, ios[k % T], 0.0025, memory=capi.MEM_DEVICE, stream=st_),
                                             max(2, min(args.warmup, 5)), regions)
            ok_all = True
            for k in range(T):   # every tick's statuses, untimed
                capi.full_tick(tctx, ios[k % T], 0.0025, memory=capi.MEM_DEVICE, stream=stream)
                torch.cuda.synchronize()
                ok_all = ok_all and bool(((keep["status"] == 0) | (keep["status"] == 4)).all().item()) and bool((keep["message_status"] == 0).all().item())
            # algorithmic bytes per robot: its message + measured state (q, qd, qd_oldest 288, base pose / twist 104, contact 4) +
            # persistent state read and written (2 x (4 + 4 + 96 + 4 + 4 + 96 + 96)) + efforts 96 + codes / statuses 12
            algo = nbytes + (288 + 104 + 4 + 2 * 304 + 96 + 12) * batch
            out["warm" if warm else "cold"] = {
                "value": batch * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3, "kernel_ms": tick_ms,
                "all_status_ok": ok_all, "roofline_frac": algo / (tick_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "warm_rejected": tctx.counter(capi.COUNTER_WARM_RETRIES)}
            tctx.close()
        e = dict(out["warm"], unit="ticks/s", batch=batch, message_bytes=nbytes // batch, states_per_replay=args.steps, ticks=T,
                 method=("qlamd_full_tick_batch (unpack + leg state machine, then balance blocks + swing-branch blocks: two launches per "
                         "tick), the balance solve warm-started from the working set the tick keeps (qlamd_tick_batch::working_set)"
                         if batch <= 16384 else
                         "qlamd_full_tick_batch (four launches per tick: unpack, leg state machine, balance solve, swing branch), the "
                         "balance solve in the placed loop the tick keeps the state of (qlamd_tick_batch::placement_state) and "
                         "warm-started from the working set it keeps (::working_set)"),
                 cold_start=out["cold"], switched_per_tick=float(np.mean(synth.support_switches(states))) if T > 1 else 0.0)
        e["kernel_ms_note"] = "both launches of a tick (HIP events around K captured ticks)"
        return e

    def wholebody_entry(batch):
        """SURVEY 8 row f4 (what the north star describes and the reference lacks): one whole-body control step per robot --
        18-DoF inverse dynamics -> force / torque QP -> 12 joint efforts -- on a trot trajectory (joints moved by their rates),
        placed and warm-started through qlamd_place_next_call (64-bit working sets, updated in place), and cold."""
        T, regions = ticks_for(200)
        states = synth.wholebody_trajectory(batch, "trot", T)
        dsw = [capi.to_device(s_, dev) for s_ in states]
        tau, grf = torch.zeros(batch, 12, dtype=torch.float64, device=dev), torch.zeros(batch, 12, dtype=torch.float64, device=dev)
        st_w = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        out = {}
        import ctypes as C
        for warm in (False, True):
            wctx = capi.Context(device=local_rank)
            orders = [torch.arange(batch, dtype=torch.int32, device=dev) for _ in range(2)]
            its = [torch.zeros(batch, dtype=torch.int32, device=dev) for _ in range(2)]
            wsets = torch.zeros(batch, 2, dtype=torch.int32, device=dev)

            def step(k, st_, wctx=wctx, orders=orders, its=its, wsets=wsets, warm=warm):
                pl = capi.Placement(orders[k & 1].data_ptr(), its[k & 1].data_ptr(), its[(k - 1) & 1].data_ptr(), orders[(k + 1) & 1].data_ptr(),
                                    int(os.environ.get("QLAMD_BENCH_WB_POLICY", capi.PLACEMENT_AUTO)) if warm else capi.PLACEMENT_AUTO,
                                    wsets.data_ptr() if warm else None, wsets.data_ptr() if warm else None)
                rc = capi.lib().qlamd_place_next_call(wctx._h, C.byref(pl))
                if rc != 0:
                    raise capi.QlamdError(rc, "qlamd_place_next_call")
                capi.wholebody_solve_device(wctx, dsw[k % T], tau, grf, st_w, stream=st_)
            elapsed, kernel_ms = time_captured(step, max(2, min(args.warmup, 5)), regions)
            ok_all = True
            for k in range(T):
                step(k, stream)
                torch.cuda.synchronize()
                ok_all = ok_all and bool((st_w == 0).all().item())
            out["warm" if warm else "cold"] = {
                "value": batch * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3, "kernel_ms": kernel_ms,
                "all_status_ok": ok_all, "roofline_frac": (272 + 52 + 196) * batch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "warm_rejected": wctx.counter(capi.COUNTER_WARM_RETRIES)}
            wctx.close()
        return dict(out["warm"], unit="whole-body control steps/s", batch=batch, states_per_replay=args.steps, ticks=T,
                    method="qlamd_wholebody_solve_batch (18-DoF inverse dynamics -> force / torque QP with friction-cone and torque-limit rows "
                           "-> 12 efforts), placed and warm-started through qlamd_place_next_call",
                    cold_start=out["cold"], switched_per_tick=float(np.mean(synth.support_switches(states))) if T > 1 else 0.0,
                    kernel_ms_note="solve launch + the placement's launches behind it (HIP events around K captured steps)")

    if args.workload in ("full_tick", "wholebody"):
        # side workloads on their own (reported separately from the headline metric): the same entries the default line carries
        # in `also`, at the batch asked for
        e = (full_tick_entry if args.workload == "full_tick" else wholebody_entry)(args.batch)
        print(json.dumps({
            "metric": ("whole control ticks/sec, message to efforts (SURVEY 8 rows a1 + f1 + f2" if args.workload == "full_tick" else
                       "whole-body control steps/sec (SURVEY 8 row f4") + ", reported separately from the headline metric)",
            "value": e["value"], "unit": e["unit"], "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": e["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": {"workload": e["method"], "robots_per_gpu": args.batch, "trot trajectory ticks": e["ticks"],
                                            "switched_per_tick": e["switched_per_tick"], "all_status_ok": e["all_status_ok"]},
            "roofline": {"bound": "hbm", "achieved": e["roofline_frac"] * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": e["roofline_frac"], "traffic": None, "kernel_ms": e["kernel_ms"], "kernel": e["kernel_ms_note"]},
            "cold_start": e["cold_start"], "warm_rejected": e["warm_rejected"]}), flush=True)
        return None

    B = args.batch
    method = args.method if not args.rpw else "plain"  # (the one-lane kernels of --rpw know neither placement nor warm start)
    others = [m for m in ("placed", "plain") if m != method] if method == "warm" else (["plain"] if method == "placed" else ["placed"])
    OTHER_KEY = {"placed": "cold_start", "plain": "unplaced", "warm": "warm_started"}
    few = min(args.replays, 5)
    res = run_preset(args.gait, args.errors, gather, args.replays, world > 1, collect=args.collect, every=args.gather_every, method=method)
    # the same preset by the other methods (every rank takes part: the preset's consensus steps are collectives)
    beside = ({m: run_preset(args.gait, args.errors, False, few, False, method=m) for m in others}
              if not (args.rpw or args.no_also) else {})
    # ... and by the line's method on records instead of per-field arrays (the optional gather-friendly layout)
    on_records = (run_preset(args.gait, args.errors, False, few, False, method=method, records=True)
                  if not (args.rpw or args.no_also or method == "plain" or collective) else None)
    # The other presets and every other BASELINE config, same process (one GPU only; fewer samples each):
    #   static-calm, trot at the headline batch; trot_b8192 (one rank's shard of configs[3]) and trot_b65536 (its global
    #   batch on one GPU), each by the line's method with the other methods beside it; pose_sqp_b4096 (configs[4]).
    also, scale_point = None, None
    SCALE_B = 8192

    def brief(r):
        return {"value": r["batch"] * args.steps / r["elapsed"], "ms_per_step": r["elapsed"] / args.steps * 1e3,
                "kernel_ms": r["kernel_ms"], "all_status_ok": r["ok"], "iterations": r.get("iterations")}

    def preset_entry(gait, errors, bb, methods_beside):
        e = entry(run_preset(gait, errors, False, few, False, batch=bb, method=method), gait, errors)
        for m in methods_beside:
            e[OTHER_KEY[m]] = brief(run_preset(gait, errors, False, few, False, batch=bb, method=m))
        return e

    if world == 1 and not collective and not args.no_also:
        also = {}
        for gait, errors in (("static", "calm"), ("static", "survey"), ("trot", "survey")):
            if gait == args.gait and (gait == "trot" or errors == args.errors):
                continue
            also["static-%s" % errors if gait == "static" else "trot"] = preset_entry(gait, errors, B, others)
        for name, bb in (("trot_b8192", 8192), ("trot_b65536", 65536)):
            if args.gait == "trot" and bb == B:
                continue
            also[name] = preset_entry("trot", "survey", bb, others)
        if not args.rpw and method != "plain" and "trot_b65536" in also:
            also["trot_b65536"]["records"] = brief(run_preset("trot", "survey", False, few, False, batch=65536, method=method, records=True))
        for name, fn in (("pose_sqp_b4096", pose_sqp_entry), ("full_tick_b4096", full_tick_entry), ("wholebody_trot_b4096", wholebody_entry)):
            try:
                also[name] = fn(4096)
            except Exception as e:  # a side measurement must not cost the line
                also[name] = {"error": repr(e)[:300]}
    # scale_point: the workload every line at every N carries, so that a weak-scaling curve can be drawn across lines:
    # 8192 trot robots per GPU (configs[3]'s shard).  efficiency(N) = scale_point(N).value / (N * scale_point(1).without_gather)
    if args.gait == "trot" and B == SCALE_B:
        sp_res = res
    elif args.no_also or (world == 1 and not collective):  # (one GPU: taken from also["trot_b8192"] below)
        sp_res = None
    else:
        sp_res = run_preset("trot", "survey", gather, few, world > 1, collect=args.collect, every=args.gather_every,
                            batch=SCALE_B, method=method)
    if rank == 0:
        if sp_res is not None:
            scale_point = {"robots_per_gpu": SCALE_B, "gait": "trot", "n_gpus": world, "method": sp_res["method"],
                           "value": world * SCALE_B * args.steps / sp_res["elapsed"],
                           "ms_per_step": sp_res["elapsed"] / args.steps * 1e3,
                           "without_gather": (world * SCALE_B * args.steps / sp_res["plain"]["elapsed"]) if sp_res.get("plain") else
                                             (world * SCALE_B * args.steps / sp_res["elapsed"] if not gather else None),
                           "result_collection": "rccl all_gather of torques" if gather else "none",
                           "trajectory": sp_res["trajectory"]}
        elif also is not None and "trot_b8192" in also:
            t8 = also["trot_b8192"]
            scale_point = {"robots_per_gpu": SCALE_B, "gait": "trot", "n_gpus": 1, "method": t8["method"], "value": t8["value"],
                           "ms_per_step": t8["ms_per_step"], "without_gather": t8["value"],
                           "result_collection": "none (one GPU: the result is already whole)",
                           "trajectory": t8["trajectory"],
                           **{OTHER_KEY[m]: t8[OTHER_KEY[m]] for m in others if OTHER_KEY[m] in t8}}
        if scale_point is not None:
            scale_point["definition"] = ("8192 trot robots per GPU (BASELINE configs[3]'s shard), same method and timed region as "
                                         "`value`; weak-scaling efficiency(N) = scale_point(N).value / (N x scale_point(1).without_gather)")

    METHOD_NOTE = {
        "warm": ("the caller's loop of include/qlamd.h: every step is ONE launch of qlamd_balance_solve_placed_batch that solves tick k "
                 "of the trajectory in the placement (which four robots share a wavefront) the previous step's launch made from the "
                 "iteration counts of the step before it, starts every robot's active-set loop from its final working set of tick "
                 "k - 1 (a robot whose warm start fails the final check is solved again cold inside the same launch: "
                 "`trajectory.warm_rejected`), writes counts and sets and makes the next step's placement with extra wavefronts of "
                 "its own -- hints, placement and solve all inside the timed region, hints always from EARLIER ticks.  The QP's "
                 "minimiser is unique: efforts are those of the cold start to the solver's accuracy and within 1e-6 of the oracle's "
                 "on every tick (tests/test_trajectory_gpu.py); iteration counts are no longer QuadProg++'s.  `cold_start` = the "
                 "same steps with every QP started from the empty working set (the placed loop, the headline of round 5), "
                 "`unplaced` = through qlamd_balance_solve_batch (robot s in slot s, the headline of rounds 1-4).  The placement's "
                 "policy is QLAMD_PLACEMENT_AUTO: with a warm start that is NO placement up to 4096 robots (the launch itself leaves "
                 "the identity as the next order: a warm-started launch lasts as long as its slowest robot, whoever its neighbours "
                 "are) and the throughput policy above (profiles/r6/ab_warm_policies.txt)"),
        "placed": ("every step is ONE launch of qlamd_balance_solve_placed_batch: it solves all robots, each from the empty working set, "
                   "in the placement that the previous step's launch made from the iteration counts of the step before it, writes its "
                   "own counts, and makes the next step's placement with extra wavefronts of its own; results bit for bit those of "
                   "the plain entry (tests/test_placement_gpu.py)"),
        "plain": "qlamd_balance_solve_batch: robot s in slot s of the launch, every QP from the empty working set"}
    line = None
    if rank == 0:
        elapsed = res["elapsed"]
        total = world * B * args.steps
        roof, valu = roofline_of(res, args.gait, args.errors)
        line = {
            "metric": "control-step QP solves/sec (18-DoF, 4-contact) at 1/2/4/8 MI355X",
            "value": total / elapsed, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "batch=%d robots per GPU, %s, one balance-controller control step "
                                   "(virtual-model wrench + leg FK + force-distribution QP + torques) per robot and step, "
                                   "the K steps of a timed region on K consecutive control ticks of the same robots"
                                   % (B, "static 4-contact stance" if args.gait == "static"
                                      else "trot gait (2<->4 contacts)"),
                       "robots_per_gpu": B, "global_batch": world * B, "gait": args.gait, "seed": synth.SEED,
                       "method": method, "method_note": METHOD_NOTE[method],
                       "states_per_replay": res["trajectory"]["states_per_replay"],   # K states solved per timed region, all different
                       "trajectory": res["trajectory"],
                       "trajectory_note": ("synth.trajectory: tick t + 1 = tick t with the measured / desired pose integrated over "
                                           "2.5 ms with the measured / desired twist and, for a trot, the gait phase advanced by "
                                           "2.5 ms / 0.9 s and the support flags recomputed (switched_per_tick: robots whose support set "
                                           "changes from one tick to the next); step k of the loop solves tick k % ticks; a timed region is "
                                           "K steps (states_per_replay), consecutive regions continue the trajectory (`regions` captured "
                                           "graphs replayed in turn), so the jump back to tick 0 (switched_at_wrap; the steps after it run "
                                           "on hints a real loop never sees that stale) falls into one region in `regions`"),
                       "iterations": res.get("iterations"),
                       "tracking_error": list(synth.tracking_error(args.gait, args.errors)),
                       "tracking_error_note": "half-widths of the uniform position (m) / rotation-vector (rad) / twist errors at tick 0: "
                                              "SURVEY.md 8(d)'s 0.02 / 0.05 / 0.1 unless --errors calm (static only; `also` "
                                              "carries the other presets)",
                       "result_collection": ("rccl all_gather of torques" if args.collect == "rccl" else
                                             "copies of the torque shard into every rank's buffer (hipMemcpyAsync, IPC-mapped)") if gather else
                       ("none (--no-gather)" if collective else "none (single GPU)"),
                       "rccl_ranks": ranks_seen if collective else None,
                       "launch": "hipGraph of K steps" if res["graph"] else "eager",
                       "gather_stream": ("second captured stream" if res["overlap"] else "solve stream") if gather else None,
                       "gather_layout_ok": res.get("gather_layout_ok"),
                       "collect": args.collect if gather else None, "gather_every": args.gather_every if gather else None,
                       "timed_region": "median of %d samples of exactly K steps, each between barrier + synchronize" % len(res["samples_ms"]),
                       "settle_ms": args.settle_ms,
                       "samples_ms": res["samples_ms"],
                       "all_status_ok": res["ok"]},
            "roofline": roof,
        }
        if res["plain"] is not None:
            line["without_gather"] = {"value": total / res["plain"]["elapsed"], "ms_per_step": res["plain"]["elapsed"] / args.steps * 1e3,
                                      "samples_ms": res["plain"]["samples_ms"]}
        if valu:
            line["valu_issue"] = valu
        for m, r in beside.items():
            line[OTHER_KEY[m]] = entry(r, args.gait, args.errors)
        if on_records is not None:
            line["records"] = dict(brief(on_records), note="the same steps with the nine double fields of a robot in one 48-double record "
                                   "(QLAMD_OPT_STATE_LAYOUT = QLAMD_STATE_RECORDS, qlamd_state_record) instead of one array per field, the "
                                   "layout of the reference's hardware interface that `value` is measured on")
        if scale_point is not None:
            line["scale_point"] = scale_point
        if also is not None:
            line["also"] = also
        if cpu is not None:
            line["cpu_baseline"] = cpu

    # The same steps with the other ways of collecting the results, so that a scaling run has something to compare the
    # per-step all-gather (the default: BASELINE's north star) with: one all-gather per 8 steps, and the peer-copy form.
    # They run after the line is complete and under a watchdog: an alternative that hangs (a collective or a mapped peer
    # buffer some rank never reaches) costs the line its `alternatives` object, not the measurement.
    emit_lock, emitted = threading.Lock(), [False]

    def emit(alternatives):
        with emit_lock:
            if emitted[0]:
                return
            emitted[0] = True
            if rank == 0:
                if alternatives is not None:
                    line["alternatives"] = alternatives
                os.write(json_fd, (json.dumps(line) + "\n").encode())

    alternatives = None
    if gather and (world > 1 or args.force_collective) and args.alternatives and not args.no_alternatives:
        def give_up():
            # A hung alternative (a collective or a mapped peer buffer some rank never reaches): the line, complete but for
            # this object, is written with the error in it.  The process cannot be unwound past a hung device call, so it
            # ends here, on every rank with exit code 3: a launcher must never read success from a rank that did not finish
            # what it was asked to do (the measurement on the line is whole all the same, the error is in `alternatives.error`).
            try:
                emit({"error": "not finished after %g s; the rest of the line is complete" % args.alternatives_timeout})
            finally:
                os._exit(3)
        guard = threading.Timer(args.alternatives_timeout, give_up)
        guard.daemon = True
        guard.start()
        alternatives = {}
        for name, (col, ev) in (("rccl_every_8", ("rccl", 8)), ("peer_every_1", ("peer", 1)), ("peer_every_8", ("peer", 8))):
            if col == args.collect and ev == args.gather_every:
                continue
            err = None
            try:
                r = run_preset(args.gait, args.errors, True, min(args.replays, 5), False, collect=col, every=ev, method=method)
                alternatives[name] = {"value": world * B * args.steps / r["elapsed"], "ms_per_step": r["elapsed"] / args.steps * 1e3,
                                      "collect": col, "gather_every": ev, "layout_ok": r.get("gather_layout_ok"),
                                      "launch": "hipGraph of K steps" if r["graph"] else "eager"}
            except Exception as e:  # an alternative that does not run costs the line nothing
                err = repr(e)[:200]
                alternatives[name] = {"error": err}
            # a failure on one rank only would leave the others in the next alternative's collectives: agree, and stop here
            bad = torch.tensor([1 if err else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()):
                if err is None:
                    alternatives[name] = {"error": "failed on another rank"}
                break
        guard.cancel()
    emit(alternatives)

    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
