#!/usr/bin/env python3
"""GPU probe: wall-clock latency of one host-buffer call (batch 1, the reference's own calling pattern, config 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quadruped_locomotion_amd import capi, synth
ctx = capi.Context()
for B in (1, 16, 256, 4096, 65536):
    s = synth.make_states(B, "trot")
    for _ in range(20):
        ctx.balance_solve_host(s)
    t = []
    for _ in range(200):
        t0 = time.perf_counter(); ctx.balance_solve_host(s); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    print("balance host call B=%d: median %.1f us, p10 %.1f, p90 %.1f  -> %.2f M steps/s PCIe-inclusive" % (B, np.median(t), np.percentile(t, 10), np.percentile(t, 90), B / np.median(t)))
