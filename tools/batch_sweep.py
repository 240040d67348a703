#!/usr/bin/env python3
"""GPU probe: step time vs batch size for the three launch geometries (sets pick_rpw thresholds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quadruped_locomotion_amd import capi, synth

def timeit(ctx, d, B, rpw, reps):
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0"); st = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    ctx.set_robots_per_wave(rpw)
    for _ in range(3): ctx.balance_solve_device(d, tau, None, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ctx.balance_solve_device(d, tau, None, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

ctx = capi.Context()
for gait in ("static", "trot"):
    for B in (1024, 4096, 8192, 16384, 32768, 65536, 262144, 1048576):
        d = capi.to_device(synth.make_states(B, gait))
        reps = 50 if B <= 65536 else 10
        t = {r: timeit(ctx, d, B, r, reps) for r in (4, 16, 64)}
        best = min(t, key=t.get)
        print("%-6s B=%8d  coop(4) %8.1f us  rpw16 %8.1f us  rpw64 %8.1f us   best=%d  %.0f M solves/s" % (
            gait, B, t[4], t[16], t[64], best, B / t[best]))
