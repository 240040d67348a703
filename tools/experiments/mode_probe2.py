"""Which buffer carries the placed loop's mode?  One set of buffers; then ONE kind of buffer at a time is replaced by a fresh
allocation (the old one kept alive) and the loop timed again (graph of 200 steps, best of 5 replays).  usage: mode_probe2.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_locomotion_amd import capi, synth

B = 4096
state = synth.make_states(B, "static", errors="survey")
ctx = capi.Context(device=0)
keep = []


def fresh(kind, b):
    keep.append(dict(b))
    if kind == "state":
        b["d"] = capi.to_device(state)
    elif kind == "tau":
        b["tau"] = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    elif kind == "status":
        b["status"] = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    elif kind == "orders":
        b["order"] = [torch.arange(B, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    elif kind == "iters":
        b["iters"] = [torch.zeros(B, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    elif kind == "pad":
        keep.append(torch.zeros(3 * 1024 * 1024 + 4096 * len(keep), dtype=torch.uint8, device="cuda:0"))


def run(b, K=200):
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
    return min(ts)


b = {}
for kind in ("state", "tau", "status", "orders", "iters"):
    fresh(kind, b)
print("first set: %.2f" % run(b))
for rnd in range(3):
    for kind in ("pad", "tau", "status", "orders", "iters", "state"):
        fresh(kind, b)
        print("round %d, fresh %-7s: %.2f   (tau %x status %x order %x %x iters %x %x q %x)" % (
            rnd, kind, run(b), b["tau"].data_ptr() & 0xFFFFFFF, b["status"].data_ptr() & 0xFFFFFFF, b["order"][0].data_ptr() & 0xFFFFFFF,
            b["order"][1].data_ptr() & 0xFFFFFFF, b["iters"][0].data_ptr() & 0xFFFFFFF, b["iters"][1].data_ptr() & 0xFFFFFFF,
            b["d"]["q"].data_ptr() & 0xFFFFFFF), flush=True)
