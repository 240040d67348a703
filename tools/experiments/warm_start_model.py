#!/usr/bin/env python3
"""Offline (CPU, numpy): what starting the dual active-set method from the PREVIOUS control step's final working set would
buy -- on the bench batches, previous step = the states one control period (2.5 ms) earlier (placement_model.earlier_state).
Procedure (active_set_paths.Solver): install the previous working set with fast adds (no step lengths, no selection), drop
negative multipliers one at a time until the pair is dual feasible, then run the method as usual.  Counts per robot and a
cost estimate with the measured pass costs (add 0.60, drop 0.50 us) and the estimated fast ones (0.25 / 0.45 us).
usage: warm_start_model.py [static|trot] [calm|survey] [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import active_set_paths as P  # noqa: E402
import placement_model as M   # noqa: E402

C_ADD, C_DROP, C_FADD, C_FDROP = 0.60, 0.50, 0.25, 0.45


def run(qp, warm_set):
    S = P.Solver(*qp)
    if warm_set is not None:
        for p in warm_set:
            if len(S.act) < S.n:
                S.fast_add(p)
        if not S.drop_negative():
            return None, S
    st = S.run("most_violated")
    return st, S


def main():
    gait = sys.argv[1] if len(sys.argv) > 1 else "static"
    err = sys.argv[2] if len(sys.argv) > 2 else "survey"
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    state = P.synth.make_states(n, gait, errors=None if gait == "trot" else err)
    prev = M.earlier_state(state, M.DT)
    cold, warm, fail, same_set, xerr = [], [], 0, 0, 0.0
    for i in range(n):
        if not state["stance"][i].any():
            continue
        _, S0 = run(P.qp_of(prev, i), None)
        qp = P.qp_of(state, i)
        st_c, Sc = run(qp, None)
        st_w, Sw = run(qp, list(S0.act))
        cold.append(C_ADD * Sc.c["add"] + C_DROP * Sc.c["drop"])
        if st_w is None or st_w != "ok":
            fail += 1
            warm.append(cold[-1])
            continue
        warm.append(C_ADD * Sw.c["add"] + C_DROP * Sw.c["drop"] + C_FADD * Sw.c["fadd"] + C_FDROP * Sw.c["fdrop"])
        same_set += sorted(S0.act) == sorted(Sc.act)
        xerr = max(xerr, float(np.abs(Sw.x - Sc.x).max()))
    cold, warm = np.array(cold), np.array(warm)
    print("%s-%s, %d robots: previous working set = this one for %.1f %%; warm start fails (falls back to cold) for %d; max |x_warm - x_cold| %.1e"
          % (gait, err, len(cold), 100.0 * same_set / len(cold), fail, xerr))
    for name, c in (("cold", cold), ("warm", warm)):
        print("  %-5s us above the floor per robot: mean %.2f  p99 %.2f  max %.2f" % (name, c.mean(), np.percentile(c, 99), c.max()))


if __name__ == "__main__":
    main()
