"""Which of the caller's buffers decides the placed loop's mode?  ALL of them (ten state arrays, efforts, status, two orders, two
count arrays) are laid into one allocation by a recipe, and the loop is timed (graph of 200 steps, best-worst of 5 replays).
usage: layout_probe2.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_locomotion_amd import capi, synth

B = 4096
state = synth.make_states(B, "static", errors="survey")
ctx = capi.Context(device=0)
keep = []
SMALL = [("status", 4 * B), ("order0", 4 * B), ("order1", 4 * B), ("iters0", 4 * B), ("iters1", 4 * B)]


def lay(recipe):
    """recipe(name, nbytes, cursor) -> offset; buffers in the order: state fields, tau, the five small arrays"""
    buf = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda:0")
    keep.append(buf)
    cur = 0
    d = {}
    def place(name, nbytes):
        nonlocal cur
        off = recipe(name, nbytes, cur)
        off = (off + 7) // 8 * 8
        cur = off + nbytes
        return buf[off:off + nbytes]
    for k, v in state.items():
        a = np.ascontiguousarray(v)
        t = place(k, a.nbytes).view(torch.float64 if a.dtype == np.float64 else torch.uint8).view(*a.shape)
        t.copy_(torch.from_numpy(a))
        d[k] = t
    tau = place("tau", 96 * B).view(torch.float64).view(B, 12)
    sm = {n: place(n, nb).view(torch.int32) for n, nb in SMALL}
    sm["order0"].copy_(torch.arange(B, dtype=torch.int32, device="cuda:0")); sm["order1"].copy_(sm["order0"])
    return d, tau, sm


def run(d, tau, sm, K=200):
    order, iters, status = [sm["order0"], sm["order1"]], [sm["iters0"], sm["iters1"]], sm["status"]
    def step(k, st):
        ctx.balance_solve_placed_device(d, tau, None, status, order=order[k & 1], iterations=iters[k & 1],
                                        prev_iterations=iters[(k - 1) & 1], next_order=order[(k + 1) & 1],
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
    return "%.2f-%.2f" % (min(ts), max(ts))


al = lambda x, a: (x + a - 1) // a * a
recipes = {
    "packed, 512-byte aligned": lambda n, nb, c: al(c, 512),
    "packed, 4 KB aligned": lambda n, nb, c: al(c, 4096),
    "packed, 64 KB aligned": lambda n, nb, c: al(c, 65536),
    "every buffer on a 2 MB boundary": lambda n, nb, c: al(c, 1 << 21),
    "2 MB boundaries + 4 KB x index": None,
    "state packed; tau and small arrays on 2 MB boundaries": lambda n, nb, c: al(c, 1 << 21) if n in ("tau",) or n in dict(SMALL) else al(c, 512),
    "state on 2 MB boundaries; tau and small arrays packed": lambda n, nb, c: al(c, 512) if n in ("tau",) or n in dict(SMALL) else al(c, 1 << 21),
    "packed, small arrays 16 KB + 256 B apart": lambda n, nb, c: al(c, 512) + (256 if n in dict(SMALL) else 0),
}
idx = [0]
def rec5(n, nb, c):
    idx[0] += 1
    return al(c, 1 << 21) + 4096 * idx[0]
recipes["2 MB boundaries + 4 KB x index"] = rec5
for rep in range(3):
    for name, r in recipes.items():
        idx[0] = 0
        d, tau, sm = lay(r)
        print("%-58s %s" % (name, run(d, tau, sm)), flush=True)
