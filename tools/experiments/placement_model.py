#!/usr/bin/env python3
"""Offline (CPU, numpy): what the launch of a 4096-robot batch costs when robots are PLACED into wavefronts by a hint
instead of by index -- on the real add / drop sequences of the bench batches (the loop of csrc/force_qp_coop.hpp restated
in active_set_paths.py, costs as in lockstep_schemes.py scheme B = what the kernel does since round 4).

A wavefront holds four robots in lockstep; its cost is the union of their passes (every pass ends with an add for all
live rows, rows that are blocked drop first inside the pass).  The launch lasts as long as its slowest wavefront, so the
question is how much of the gap between "slowest wavefront" and "slowest robot alone" a placement buys back, and how good
the hint has to be.

hints (one integer per robot, larger = harder):
  passes      adds + drops of THIS solve (upper bound: a perfect hint)
  adds        outer iterations of this solve only (what QuadProg++ calls iter)
  prev_tick   passes of the solve of the state one control period (2.5 ms, balance_controller_manager.cpp:48) EARLIER:
              the measured state integrated backwards with its own twist, the desired state with the desired twist --
              what a 400 Hz caller has for free from its last tick
  viol_x0     number of rows violated at the unconstrained minimiser (available inside the kernel before the loop)
placements (slot s = row s % 4 of wavefront s // 4):
  identity    robot i in slot i
  snake       robots sorted by hint, hardest first; rank k goes to wavefront k (k < W), then W-1-(k-W), ... (boustrophedon:
              the hardest W robots one per wavefront, each joined by the easiest of the next tiers)
  top<p>      the hardest p % of the robots each share a wavefront with the three easiest left; the others by index
  sorted      robots sorted by hint, four neighbours per wavefront (the wrong way round, for contrast)
usage: placement_model.py [static|trot] [calm|survey]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import active_set_paths as P  # noqa: E402
import lockstep_schemes as L  # noqa: E402

# scheme-B costs calibrated on round 4's probes: an add pass 0.60 us; a round of drops 0.50 us when every live row drops
# (a lone robot with ghost rows: profiles/r4/row_mix_probe.txt, robot 128: 14.35 us above the floor against 14.2 here) and
# 0.68 us when live rows wait under an execution mask (wavefront 830: 18.45 measured)
C_PASS, C_DROP_ALL, C_DROP_MASKED = 0.60, 0.50, 0.68
FLOOR, RAMP = 4.55, 0.65          # a wavefront without a pass; launch minus its slowest wavefront (wave_scan.txt)
DT = 0.0025


def wave_cost(seqs):
    its = [L.iterations(s) for s in seqs]
    total = 0.0
    for j in range(max((len(i) for i in its), default=0)):
        live = [i[j] for i in its if j < len(i)]
        rounds = max(live)
        total += C_PASS
        for rnd in range(rounds):
            total += C_DROP_ALL if all(d > rnd for d in live) else C_DROP_MASKED
    return total


def earlier_state(state, dt):
    """The state one control period earlier, to first order in dt."""
    s = {k: v.copy() for k, v in state.items()}
    s["base_pos"] = state["base_pos"] - dt * state["base_linvel"]
    s["des_pos"] = state["des_pos"] - dt * state["des_linvel"]

    def back(quat, omega_world):
        # q(t - dt) = exp(-dt omega) * q(t)
        rv = -dt * omega_world
        return P.synth._quat_mul(P.synth._quat_exp(rv), quat)
    # base_angvel is expressed in the base frame (VirtualModelController.cpp:150-151): rotate to world
    Rw = np.stack([P.O.quat_to_matrix(q) for q in state["base_quat"]])
    s["base_quat"] = back(state["base_quat"], np.einsum("bij,bj->bi", Rw, state["base_angvel"]))
    s["des_quat"] = back(state["des_quat"], np.einsum("bij,bj->bi", Rw, state["des_angvel"]))
    return s


def viol_x0(qp):
    S = P.Solver(*qp)
    return int((S.slacks() < 0.0).sum())


def place(kind, hint, n):
    W = n // 4
    order = np.argsort(-hint, kind="stable")          # hardest first, ties by index
    if kind == "identity":
        return np.arange(n)
    if kind == "sorted":
        return order
    if kind == "snake":
        slots = np.empty(n, dtype=np.int64)
        for k, robot in enumerate(order):
            tier, pos = divmod(k, W)
            w = pos if tier % 2 == 0 else W - 1 - pos
            slots[4 * w + tier] = robot
        return slots
    if kind.startswith("top"):
        p = float(kind[3:]) / 100.0
        nh = int(round(p * n))
        hard, easy = order[:nh], order[::-1][:3 * nh]
        taken = set(hard.tolist()) | set(easy.tolist())
        rest = [i for i in range(n) if i not in taken]
        slots = []
        for k in range(nh):
            slots += [hard[k], easy[3 * k], easy[3 * k + 1], easy[3 * k + 2]]
        return np.array(slots + rest)
    raise ValueError(kind)


def main():
    gait = sys.argv[1] if len(sys.argv) > 1 else "static"
    err = sys.argv[2] if len(sys.argv) > 2 else "survey"
    n = 4096
    state = P.synth.make_states(n, gait, errors=None if gait == "trot" else err)
    qps = [P.qp_of(state, i) for i in range(n)]
    seqs = [L.sequence(qp) for qp in qps]
    prev = earlier_state(state, DT)
    seqs_prev = [L.sequence(P.qp_of(prev, i)) for i in range(n)]
    hints = {
        "passes": np.array([len(s) for s in seqs]),
        "adds": np.array([s.count("a") for s in seqs]),
        "prev_tick": np.array([len(s) for s in seqs_prev]),
        "viol_x0": np.array([viol_x0(qp) for qp in qps]),
    }
    lone = np.array([wave_cost([s]) for s in seqs])
    print("%s-%s, %d robots: passes mean %.1f max %d; slowest robot alone %.2f us above the floor (robot %d)"
          % (gait, err, n, hints["passes"].mean(), hints["passes"].max(), lone.max(), int(lone.argmax())))
    print("  hint quality (rank correlation with the true passes): " + "  ".join(
        "%s %.3f" % (k, np.corrcoef(np.argsort(np.argsort(v)), np.argsort(np.argsort(hints["passes"])))[0, 1])
        for k, v in hints.items()))
    same = (hints["prev_tick"] == hints["passes"]).mean()
    print("  prev_tick: same pass count as this tick for %.1f %% of the robots, |difference| mean %.2f max %d"
          % (100 * same, np.abs(hints["prev_tick"] - hints["passes"]).mean(), np.abs(hints["prev_tick"] - hints["passes"]).max()))
    print("  %-10s %-9s  slowest wavefront  mean   -> launch estimate (floor %.2f + ramp %.2f)" % ("placement", "hint", FLOOR, RAMP))
    for kind in ("identity", "snake", "top5", "top10", "top25", "sorted"):
        for hname, h in hints.items():
            if kind == "identity" and hname != "passes":
                continue
            slots = place(kind, h, n)
            assert sorted(slots.tolist()) == list(range(n))
            c = np.array([wave_cost([seqs[r] for r in slots[w:w + 4]]) for w in range(0, n, 4)])
            print("  %-10s %-9s  %6.2f            %6.2f -> %.2f us" % (kind, "-" if kind == "identity" else hname, c.max(), c.mean(), FLOOR + c.max() + RAMP))


if __name__ == "__main__":
    main()
