import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quadruped_locomotion_amd import capi, synth
# diagnostic build: python -c "from quadruped_locomotion_amd import build; build.build(defines=('QLAMD_STAMPS',), lib='variants/libqlamd_stamps.so')"
capi.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "variants", "libqlamd_stamps.so")
from oracle import oracle as O
s = synth.make_states(4096, "trot")
its = np.array([O.balance_step(s, i)["iters"] for i in range(4096)])
names = ["load", "wrench", "FK/J/G", "pyramid", "G+GaussJordan", "x0", "loop", "refine", "torque+store"]
ctx = capi.Context(); ctx.set_robots_per_wave(4)
for target in (1, 4, 8, 11):
    idx = np.where(its == target)[0][:1]
    rep = {k: np.repeat(v[idx], 4096, axis=0) for k, v in s.items()}
    d = capi.to_device(rep)
    tau = torch.zeros(4096, 12, dtype=torch.float64, device="cuda:0"); st = torch.zeros(4096, dtype=torch.int32, device="cuda:0")
    for _ in range(3): ctx.balance_solve_device(d, tau, None, st)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    capi.lib().qlamd_debug_stamps(out, 16)
    t = np.array(out[:10], dtype=np.float64)
    print("iters", target, "total ticks", t[9] - t[0], "(100 MHz ticks?)")
    t = np.array(out[:11], dtype=np.float64)
    for k in range(9): print("   %-14s %8.0f" % (names[k], t[k + 1] - t[k]))
    print("   empty segment (stamp overhead) %.0f" % (t[10] - t[9]))
