#!/usr/bin/env python3
"""Generate tests/golden/force_qp_exact.npz: 40-digit minimisers of the golden force-distribution QPs.

The reference solves these QPs through ooqpei / OOQP (ContactForceDistribution.cpp:490), which is absent here and
un-vendored: parity at that boundary cannot be pinned.  Each QP is strictly convex (W = 1e-4 I), so the target is its
unique minimiser; this script computes it to 40 significant digits, independently of every solver in the repo:
  * the problem data are the doubles of tests/golden/qp_goldens.npz, taken as exact rationals;
  * a primal active-set iteration in mpmath arithmetic (mp.dps = 50) on the KKT system
        [ G  -N ] [x]   [-g0 ]
        [ N'  0 ] [u] = [-ci0]        N = columns of CI in the working set
    started from the working set the stored solution suggests (slack < 1e-7); a negative multiplier leaves, the most
    violated row enters, until u >= 0 and every slack >= 0: with G positive definite that certifies THE minimiser;
  * stored: x rounded to double (x), the rounding remainder (x_lo, so that x + x_lo carries ~32 digits), the
    multipliers, the working set and the worst KKT residual seen in 50-digit arithmetic.
Runs anywhere (no reference needed): the inputs are the committed goldens.
"""
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mp.mp.dps = 50


def solve_exact(G, g0, CI, ci0, x_hint):
    n, m = CI.shape
    Gm = mp.matrix(G.tolist())
    g = mp.matrix(g0.tolist())
    C = mp.matrix(CI.tolist())
    c0 = mp.matrix(ci0.tolist())
    slack_hint = CI.T @ x_hint + ci0
    W = [k for k in range(m) if slack_hint[k] < 1e-7]
    for _ in range(200):
        q = len(W)
        K = mp.zeros(n + q, n + q)
        rhs = mp.zeros(n + q, 1)
        for i in range(n):
            for j in range(n):
                K[i, j] = Gm[i, j]
            rhs[i] = -g[i]
        for a, k in enumerate(W):
            for i in range(n):
                K[i, n + a] = -C[i, k]
                K[n + a, i] = C[i, k]
            rhs[n + a] = -c0[k]
        sol = mp.lu_solve(K, rhs)
        x = sol[:n]
        u = sol[n:]
        if q and min(u) < 0:
            W.pop(int(np.argmin([float(v) for v in u])))
            continue
        slack = [sum(C[i, k] * x[i] for i in range(n)) + c0[k] for k in range(m)]
        worst = min(range(m), key=lambda k: slack[k])
        if slack[worst] < -mp.mpf(10) ** (-40) and worst not in W:
            W.append(worst)
            continue
        # KKT residual: stationarity G x + g0 - N u, in 50-digit arithmetic
        res = max(abs(sum(Gm[i, j] * x[j] for j in range(n)) + g[i] - sum(C[i, k] * u[a] for a, k in enumerate(W))) for i in range(n))
        return x, u, W, res
    raise RuntimeError("no convergence")


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "qp_goldens.npz"))
    out = {}
    for name in ("n12", "n6"):
        G, g0, CI, ci0, xr = g[name + "_G"], g[name + "_g0"], g[name + "_CI"], g[name + "_ci0"], g[name + "_x"]
        B, n = g0.shape
        m = ci0.shape[1]
        x, x_lo, u = np.zeros((B, n)), np.zeros((B, n)), np.zeros((B, m))
        active = np.zeros((B, m), np.uint8)
        res = np.zeros(B)
        for b in range(B):
            xe, ue, W, r = solve_exact(G[b], g0[b], CI[b], ci0[b], xr[b])
            for i in range(n):
                x[b, i] = float(xe[i])
                x_lo[b, i] = float(xe[i] - mp.mpf(x[b, i]))
            for a, k in enumerate(W):
                u[b, k] = float(ue[a])
                active[b, k] = 1
            res[b] = float(r)
        out.update({name + "_x": x, name + "_x_lo": x_lo, name + "_u": u, name + "_active": active, name + "_kkt_residual": res})
        print(name, "max |x_exact - x_reference_quadprog| = %.3e" % np.abs(x - xr).max(), " worst 50-digit KKT residual %.1e" % res.max(),
              " active rows: mean %.2f max %d" % (active.sum(1).mean(), active.sum(1).max()))
    dst = os.path.join(ROOT, "tests", "golden", "force_qp_exact.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst)


if __name__ == "__main__":
    main()
