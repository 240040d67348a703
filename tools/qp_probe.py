#!/usr/bin/env python3
"""GPU probe: qlamd_qp_solve_batch on 4096 force QPs (the golden n=12 / n=6 instances tiled), for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quadruped_locomotion_amd import capi
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "qp_goldens.npz"))
ctx = capi.Context()
for tag in ("n12", "n6"):
    rep = 4096 // 128
    G, g0, CI, ci0 = (np.tile(g[f"{tag}_{k}"], (rep,) + (1,) * (g[f"{tag}_{k}"].ndim - 1)) for k in ("G", "g0", "CI", "ci0"))
    for _ in range(4):
        x, f, st = capi.qp_solve(ctx, G, g0, None, None, CI, ci0)
    ok = g[f"{tag}_status"] == 0
    print(tag, "max |x - golden|", np.abs(x[:128][ok] - g[f"{tag}_x"][ok]).max())
