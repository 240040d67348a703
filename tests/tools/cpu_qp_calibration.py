#!/usr/bin/env python3
"""CPU calibration: the reference's own compiled QuadProg++ (oracle/_ref, built from
/root/reference where it lies -- build container only) against the oracle's plain-C restatement on the committed
force-QP goldens, single thread, QP only, the loop in C on both sides.  Relates the `cpu_baseline` of bench.py (the
restatement) to real reference code."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

g = np.load(os.path.join(ROOT, "tests", "golden", "qp_goldens.npz"))
ref = O.ref_lib()
assert ref is not None, "needs oracle/_ref/libquadprog_ref.so (make -C oracle ref, build container only)"
dp = C.POINTER(C.c_double)
for tag, label in (("n12", "4 stance legs: n=12, m=20"), ("n6", "2 stance legs: n=6, m=10")):
    G, g0, CI, ci0 = (np.ascontiguousarray(g[f"{tag}_{k}"], dtype=np.float64) for k in ("G", "g0", "CI", "ci0"))
    K, n = G.shape[0], G.shape[1]
    m = CI.shape[2]
    xr, xo = np.zeros((K, n)), np.zeros((K, n))
    args = lambda x: (n, 0, m, K, 0, G.ctypes.data_as(dp), g0.ctypes.data_as(dp), CI.ctypes.data_as(dp), ci0.ctypes.data_as(dp), x.ctypes.data_as(dp))  # noqa: E731
    rate = {}
    for name, fn, x in (("reference", ref.ref_solve_quadprog_batch, xr), ("restatement", O.lib().oracle_solve_quadprog_batch, xo)):
        a = list(args(x))
        a[4] = 20; fn(*a)                      # warm up
        a[4] = 400
        t0 = time.perf_counter(); bad = fn(*a); dt = time.perf_counter() - t0
        assert bad == 0
        rate[name] = K * 400 / dt
    assert np.array_equal(xr, xo)              # the restatement is bit-identical to the reference solver
    print("%s: reference QuadProg++ %.0f solves/s, oracle restatement %.0f solves/s, ratio %.2f (single thread, %d goldens x 400)"
          % (label, rate["reference"], rate["restatement"], rate["restatement"] / rate["reference"], K))
