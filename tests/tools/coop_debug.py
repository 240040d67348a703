import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quadruped_locomotion_amd import capi, synth
from oracle import oracle as O
ctx = capi.Context()
for gait in ("static", "trot"):
    s = synth.make_states(4096, gait)
    d = capi.to_device(s)
    out = {}
    for rpw in (4, 16):
        ctx.set_robots_per_wave(rpw)
        tau = torch.zeros(4096, 12, dtype=torch.float64, device="cuda:0"); grf = torch.zeros_like(tau)
        st = torch.zeros(4096, dtype=torch.int32, device="cuda:0")
        ctx.balance_solve_device(d, tau, grf, st); torch.cuda.synchronize()
        out[rpw] = (tau.cpu().numpy(), grf.cpu().numpy(), st.cpu().numpy())
    t0, g0, s0 = O.balance_batch(s, nthreads=8)
    for rpw in (4, 16):
        e = np.abs(out[rpw][1] - g0).max(axis=1)
        print(gait, "rpw", rpw, "grf err pct 50/90/99/100:", np.percentile(e, [50, 90, 99, 100]), "status", np.bincount(out[rpw][2]))
    e = np.abs(out[4][1] - g0).max(axis=1)
    for i in np.argsort(-e)[:6]:
        r = O.balance_step(s, i)
        print("  robot", i, "err", e[i], "oracle iters", r["iters"], "nact", r["n_active"], "nS", int(s["stance"][i].sum()))
