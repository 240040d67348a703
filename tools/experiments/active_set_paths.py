#!/usr/bin/env python3
"""Offline experiment (CPU, numpy): length of the dual active-set path of the contact-force QP on the bench batches
under (a) different entering rules and (b) a predicted starting working set.  The minimiser is unique, so every variant
ends at the same forces (checked against the oracle's solve); what differs is the number of add / drop passes, and the
launch at 4096 robots lasts as long as its slowest wavefront (4 consecutive robots in lockstep).

The loop follows the kernel (csrc/force_qp_coop.hpp), i.e. QuadProg++.cc:216-445 with the explicit operators H, N*.
usage: active_set_paths.py [static|trot] [calm|survey] [nrobots] [first]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from quadruped_locomotion_amd import synth  # noqa: E402

EPS = 2.220446049250313e-16
C_ADD, C_DROP = 0.52, 0.75                    # measured, tools/tail_probe.py
C_FADD, C_FDROP = 0.25, 0.45                  # estimated: add / drop without step lengths and selection


def qp_of(state, i):
    q = state["q"][i].reshape(4, 3)
    stance = state["stance"][i].astype(bool)
    legs = [l for l in range(4) if stance[l]]
    Rm = O.quat_to_matrix(state["base_quat"][i])
    r = np.array([O.leg_fk(l, q[l])[0] for l in legs])
    nb = Rm.T @ (Rm @ np.array([0.0, 0.0, 1.0]))
    yB = Rm.T @ np.array([0.0, 1.0, 0.0])
    t1 = np.cross(nb, yB); t1 /= np.linalg.norm(t1)
    t2 = np.cross(nb, t1); t2 /= np.linalg.norm(t2)
    w = O.virtual_wrench(state, i)
    nS = len(legs)
    return O.force_qp_assemble(r, w, np.tile(nb, (nS, 1)), np.tile(t1, (nS, 1)), np.tile(t2, (nS, 1)))


class Solver:
    def __init__(self, G, g0, CI, ci0):
        self.G, self.g0, self.CI, self.ci0 = G, g0, CI, ci0
        self.n, self.m = CI.shape
        self.H = np.linalg.inv(G)
        L = np.linalg.cholesky(G)
        self.psi_tol = self.m * EPS * np.trace(G) * np.sum(1.0 / np.diag(L)) * 100.0
        self.Ns = np.zeros((0, self.n))
        self.act, self.u = [], np.zeros(0)
        self.x = -self.H @ g0
        self.rnorm2 = 1.0
        self.c = dict(add=0, drop=0, fadd=0, fdrop=0)

    def slacks(self):
        return self.CI.T @ self.x + self.ci0

    # ---- operator updates
    def _add_ops(self, ip, z, r, zn):
        self.H = self.H - np.outer(z, z) / zn
        self.Ns = np.vstack([self.Ns - np.outer(r, z) / zn, z / zn])
        self.act.append(ip)
        self.rnorm2 = max(self.rnorm2, zn)

    def _drop_ops(self, k):
        nt = self.Ns[k]
        Gn = self.G @ nt
        e = nt @ Gn
        coef = self.Ns @ Gn
        self.H = self.H + np.outer(nt, nt) / e
        self.Ns = np.delete(self.Ns - np.outer(coef, nt) / e, k, 0)
        self.act.pop(k)
        return nt, coef, e

    # ---- warm start pieces
    def fast_add(self, p):
        npv = self.CI[:, p]
        z = self.H @ npv
        zn = z @ npv
        if not zn > 1e-10:          # dependent on the working set (or exhausted null space)
            return False
        r = self.Ns @ npv
        t = -(npv @ self.x + self.ci0[p]) / zn
        self.x = self.x + t * z
        self.u = np.append(self.u - t * r, t)
        self._add_ops(p, z, r, zn)
        self.c["fadd"] += 1
        return True

    def fast_drop(self, k):
        uk = self.u[k]
        nt, coef, e = self._drop_ops(k)
        self.x = self.x - nt / e * uk
        self.u = np.delete(self.u - coef / e * uk, k)
        self.c["fdrop"] += 1

    def drop_negative(self):
        for _ in range(40):
            if not (len(self.u) and self.u.min() < 0.0):
                return True
            self.fast_drop(int(np.argmin(self.u)))
        return False

    # ---- the dual method from the current S-pair
    def run(self, rule="most_violated", max_outer=60):
        iters = 0
        excl = set()
        ip = -1
        for _guard in range(500):
            if ip < 0:
                iters += 1
                s = self.slacks()
                cand = [k for k in range(self.m) if k not in self.act and k not in excl and s[k] < 0.0]
                if not cand:
                    return "ok"
                psi = sum(min(0.0, s[k]) for k in range(self.m))
                if abs(psi) <= self.psi_tol:
                    return "ok"
                if iters > max_outer:
                    return "maxiter"
                if rule == "most_violated":
                    ip = min(cand, key=lambda k: s[k])
                elif rule == "steepest":      # largest gain of the dual objective for a full step: s^2 / n'Hn
                    ip = max(cand, key=lambda k: s[k] ** 2 / max(self.CI[:, k] @ self.H @ self.CI[:, k], 1e-300))
                elif rule == "longest_step":
                    ip = max(cand, key=lambda k: -s[k] / max(self.CI[:, k] @ self.H @ self.CI[:, k], 1e-300))
                elif rule == "min_force_first":
                    nS = self.m // 5
                    mf = [k for k in cand if k < nS]
                    ip = min(mf or cand, key=lambda k: s[k])
                sp, uc = s[ip], 0.0
            npv = self.CI[:, ip]
            z = self.H @ npv
            r = self.Ns @ npv
            zn = z @ npv
            t1, kdrop = np.inf, -1
            for k in range(len(self.act)):
                if r[k] > 0 and self.u[k] / r[k] < t1:
                    t1, kdrop = self.u[k] / r[k], k
            exhausted = len(self.act) >= self.n
            t2 = -sp / zn if (not exhausted and z @ z > EPS and not (-sp / zn < 0)) else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return "infeasible"
            full = np.isfinite(t2) and t2 <= t1
            if full and not zn > EPS * EPS * self.rnorm2:       # degenerate: ban the row until the next add
                excl.add(ip)
                ip = -1
                iters -= 1
                continue
            if np.isfinite(t2):
                self.x = self.x + t * z
                sp = sp + t * zn
            self.u = self.u - t * r
            uc += t
            if full:
                self.u = np.append(self.u, uc)
                self._add_ops(ip, z, r, zn)
                self.c["add"] += 1
                excl = set()
                ip = -1
            else:
                self._drop_ops(kdrop)
                self.u = np.delete(self.u, kdrop)
                self.c["drop"] += 1
        return "guard"


def predicted(S, mode):
    """rows to start from, judged at the unconstrained minimiser; at most one of a +- friction pair per leg."""
    s = S.slacks()
    nS = S.m // 5
    x = S.x.reshape(nS, 3)
    out = []
    for leg in range(nS):
        fr = [nS + 4 * leg + k for k in range(4)]
        sm = s[leg]
        sf = s[fr].copy()
        if mode == "violated_fmin_lift" and sm < 0.0:
            # with the normal force lifted to f_min the friction slacks grow by mu * (f_min - n'f) = -mu * sm
            nrm = S.CI[3 * leg:3 * leg + 3, leg]
            mu = (S.CI[3 * leg:3 * leg + 3, fr[0]] + S.CI[3 * leg:3 * leg + 3, fr[1]]) @ nrm / 2.0
            sf = sf - mu * sm
        if sm < 0.0:
            out.append(leg)
        for a, b in ((0, 1), (2, 3)):
            k = a if sf[a] <= sf[b] else b
            if sf[k] < 0.0 and fr[a] not in S.act and fr[b] not in S.act:
                out.append(fr[k])
    return out


def solve(qp, variant):
    S = Solver(*qp)
    rule = "most_violated"
    if variant.startswith("rule:"):
        rule = variant[5:]
    elif variant.startswith("warm"):
        # warm<R>:<mode>: R rounds of {add every predicted row, drop negative multipliers one at a time}
        rounds = int(variant[4]) if variant[4].isdigit() else 1
        for _ in range(rounds):
            new = [p for p in predicted(S, variant.split(":")[1]) if p not in S.act]
            if not new:
                break
            for p in new:
                if len(S.act) < S.n:
                    S.fast_add(p)
            if not S.drop_negative():
                return None, S.c, "warmfail"
    st = S.run(rule)
    return S.x, S.c, st


def cost(c):
    return C_ADD * c["add"] + C_DROP * c["drop"] + C_FADD * c["fadd"] + C_FDROP * c["fdrop"]


def main():
    gait = sys.argv[1] if len(sys.argv) > 1 else "static"
    err = sys.argv[2] if len(sys.argv) > 2 else "calm"
    nrob = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    state = synth.make_states(4096, gait, errors=None if gait == "trot" else err)
    qps = [qp_of(state, i) for i in range(first, first + nrob)]
    ref = [O.solve_quadprog(G, g0, None, None, CI, ci0) for (G, g0, CI, ci0) in qps]
    print("oracle: iterations mean %.2f max %d; status counts %s"
          % (np.mean([r["iters"] for r in ref]), max(r["iters"] for r in ref),
             dict(zip(*np.unique([r["status"] for r in ref], return_counts=True)))))
    for variant in ("rule:most_violated",
                    "warm:violated", "warm2:violated", "warm3:violated"):
        res = [solve(qp, variant) for qp in qps]
        cs = [r[1] for r in res]
        us = np.array([cost(c) for c in cs])
        wave = np.array([cost({k: max(c[k] for c in cs[w:w + 4]) for k in cs[0]}) for w in range(0, nrob, 4)])
        dev = [np.abs(r[0] - o["x"]).max() for r, o in zip(res, ref) if r[2] == "ok" and o["status"] == 0]
        sts = dict(zip(*np.unique([r[2] for r in res], return_counts=True)))
        tot = {k: sum(c[k] for c in cs) / nrob for k in cs[0]}
        worst = int(np.argmax(us))
        print("%-26s %s | stream us mean %.2f p99 %.2f max %.2f | wavefront mean %.2f p99 %.2f max %.2f | max |dx| %.1e %s | worst %d %s"
              % (variant, " ".join("%s %.2f" % kv for kv in tot.items()), us.mean(), np.percentile(us, 99), us.max(),
                 wave.mean(), np.percentile(wave, 99), wave.max(), max(dev or [0.0]), sts, first + worst, cs[worst]))


if __name__ == "__main__":
    main()
