"""Stress of the shadow wavefronts' barrier (csrc/balance_kernel.hip, placement_wave): thousands of placed launches back to back,
batch sizes and policies changing from launch to launch (1 to 64 shadow wavefronts), some launches with the wait switched off
(QLAMD_OPT_PLACEMENT_WAIT = 0: they give up and must leave the identity), eager and as replayed hipGraphs -- after every
launch next_robot_order must be a permutation and equal the documented placement (or the identity after a give-up).
usage: stress_shadow_barrier.py [launches, default 3000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quadruped_locomotion_amd import capi, synth
from test_placement_gpu import reference_placement

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ctx = capi.Context(device=0)
rng = np.random.default_rng(7)
sizes = [70, 1023, 1025, 2048, 4096, 4099, 8192, 12000, 16384, 30000, 65536, 200000]
cache = {}
bad = gave_up = 0
stream = torch.cuda.current_stream().cuda_stream
for it in range(N):
    B = sizes[rng.integers(len(sizes))]
    if B not in cache:
        s = synth.make_states(B, "trot")
        cache[B] = (s, capi.to_device(s), torch.zeros(B, 12, dtype=torch.float64, device="cuda:0"), torch.zeros(B, dtype=torch.int32, device="cuda:0"),
                    torch.zeros(B, dtype=torch.int32, device="cuda:0"), torch.zeros(B, dtype=torch.int32, device="cuda:0"))
    s, d, tau, status, iters, nxt = cache[B]
    prev = rng.integers(0, 30, size=B).astype(np.int32)
    d_prev = torch.from_numpy(prev).to("cuda:0")
    policy = [capi.PLACEMENT_LATENCY, capi.PLACEMENT_THROUGHPUT, capi.PLACEMENT_AUTO][rng.integers(3)]
    nowait = rng.random() < 0.15
    if nowait:
        ctx.set_option(capi.OPT_PLACEMENT_WAIT, 0)
    nxt.fill_(-1)
    if rng.random() < 0.3:  # as a graph replayed three times (the barrier resets itself)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                ctx.balance_solve_placed_device(d, tau, None, status, iterations=iters, prev_iterations=d_prev, next_order=nxt,
                                                policy=policy, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            g.replay()
    else:
        ctx.balance_solve_placed_device(d, tau, None, status, iterations=iters, prev_iterations=d_prev, next_order=nxt, policy=policy, stream=stream)
    torch.cuda.synchronize()
    if nowait:
        ctx.set_option(capi.OPT_PLACEMENT_WAIT, 1 << 24)
    got = nxt.cpu().numpy()
    thr = policy == capi.PLACEMENT_THROUGHPUT or (policy == capi.PLACEMENT_AUTO and B >= 16384)
    want = reference_placement(prev, thr, support=s["stance"])
    ident = np.array_equal(got, np.arange(B, dtype=np.int32))
    if ident and not np.array_equal(got, want):
        gave_up += 1
        if not nowait:
            bad += 1
            print("launch %d: B %d gave up although it was to wait" % (it, B))
    elif not np.array_equal(got, want):
        bad += 1
        print("launch %d: B %d policy %d nowait %s: wrong placement (a permutation: %s)" % (it, B, policy, nowait, len(set(got.tolist())) == B))
print("%d launches, %d wrong, %d gave up (all of them asked to)" % (N, bad, gave_up))
sys.exit(1 if bad else 0)
