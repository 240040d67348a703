"""The placed loop's two modes (21.75 / 22.4 us per step at 4096 static robots): per process, per allocation, per capture?
One process: the loop is set up several times over -- fresh buffers every time (the old ones kept alive, so addresses move),
then the same buffers captured again -- and timed (graph of 200 steps, best of 5 replays).  usage: mode_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_locomotion_amd import capi, synth

B = 4096
state = synth.make_states(B, "static", errors="survey")
keep = []


def setup(ctx):
    d = capi.to_device(state)
    bufs = dict(d=d, tau=torch.zeros(B, 12, dtype=torch.float64, device="cuda:0"), status=torch.zeros(B, dtype=torch.int32, device="cuda:0"),
                order=[torch.arange(B, dtype=torch.int32, device="cuda:0") for _ in range(2)],
                iters=[torch.zeros(B, dtype=torch.int32, device="cuda:0") for _ in range(2)])
    keep.append(bufs)
    return bufs


def run(ctx, b, K=200):
    def step(k, st):
        ctx.balance_solve_placed_device(b["d"], b["tau"], None, b["status"], order=b["order"][k & 1], iterations=b["iters"][k & 1],
                                        prev_iterations=b["iters"][(k - 1) & 1], next_order=b["order"][(k + 1) & 1],
                                        policy=capi.PLACEMENT_AUTO, stream=st)
    for k in range(10):
        step(k, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for k in range(K):
                step(k, torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / K * 1e3)
    return min(ts), max(ts)


ctx = capi.Context(device=0)
out = []
for rep in range(6):
    b = setup(ctx)
    lo, hi = run(ctx, b)
    lo2, hi2 = run(ctx, b)
    out.append("fresh buffers %.2f-%.2f, captured again %.2f-%.2f" % (lo, hi, lo2, hi2))
    if rep == 2:
        keep.append(torch.zeros(37 * 1024 * 1024 // 8, dtype=torch.float64, device="cuda:0"))  # shift the allocator
print("one context: " + " | ".join(out))
out = []
for rep in range(3):
    c2 = capi.Context(device=0)
    b = setup(c2)
    out.append("%.2f-%.2f" % run(c2, b))
    c2.close()
print("a context each: " + " | ".join(out))
