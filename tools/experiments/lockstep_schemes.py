#!/usr/bin/env python3
"""Offline (CPU, numpy): what the lockstep of four robots per wavefront costs under different ways of lining their passes
up, on the real add / drop sequences of a bench batch (the loop of csrc/force_qp_coop.hpp restated in
active_set_paths.py).  Costs in microseconds from tools/experiments/mixed_pass_probe.py and the instruction counts of
tools/kernel_isa.py: directions + step lengths 0.28, add tail 0.32, drop tail 0.35, step lengths alone 0.15.
  A  pass-aligned (what the kernel does): pass t holds step t of every live robot; all add 0.60, all drop 0.55 (the
     directions of a continued candidate are free), mixed 0.28 + 0.35 + 0.32 + 0.08
  B  iteration-aligned: every pass ends with an add for every live robot; robots that must drop first do so inside the pass
     (drop tail + step lengths from the continued directions, repeated while any robot still drops)
usage: lockstep_schemes.py [static|trot] [calm|survey]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import active_set_paths as P  # noqa: E402

C_DS, C_ADD, C_DROP, C_STEPS, C_MIX_EXTRA = 0.28, 0.32, 0.35, 0.15, 0.08


def sequence(qp):
    """'a' / 'd' string of one robot (the kernel's pivot rule)."""
    S = P.Solver(*qp)
    seq = []
    orig_add, orig_drop = S._add_ops, S._drop_ops

    def add(*a):
        seq.append("a")
        return orig_add(*a)

    def drop(*a):
        seq.append("d")
        return orig_drop(*a)
    S._add_ops, S._drop_ops = add, drop
    S.run("most_violated")
    return "".join(seq)


def cost_a(seqs):
    t, total = 0, 0.0
    while True:
        kinds = {s[t] for s in seqs if t < len(s)}
        if not kinds:
            return total
        if kinds == {"a"}:
            total += C_DS + C_ADD
        elif kinds == {"d"}:
            prev_all_drop = t > 0 and {s[t - 1] for s in seqs if t < len(s)} == {"d"}
            total += (C_STEPS if prev_all_drop else C_DS) + C_DROP
        else:
            total += C_DS + C_DROP + C_ADD + C_MIX_EXTRA
        t += 1


def iterations(s):
    """split 'aadada' into iterations ['a', 'a', 'da', 'da'] -> drops per iteration"""
    out, d = [], 0
    for ch in s:
        if ch == "d":
            d += 1
        else:
            out.append(d)
            d = 0
    if d:
        out.append(d)  # a trailing run of drops (infeasible / dual steps): its own pass
    return out


def cost_b(seqs):
    its = [iterations(s) for s in seqs]
    total = 0.0
    for j in range(max((len(i) for i in its), default=0)):
        drops = max((i[j] for i in its if j < len(i)), default=0)
        total += C_DS + C_ADD + drops * (C_DROP + C_STEPS)
    return total


def main():
    gait = sys.argv[1] if len(sys.argv) > 1 else "static"
    err = sys.argv[2] if len(sys.argv) > 2 else "survey"
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    state = P.synth.make_states(4096, gait, errors=None if gait == "trot" else err)
    seqs = [sequence(P.qp_of(state, i)) for i in range(n)]
    a = np.array([cost_a(seqs[w:w + 4]) for w in range(0, n, 4)])
    b = np.array([cost_b(seqs[w:w + 4]) for w in range(0, n, 4)])
    lone = np.array([cost_a([s]) for s in seqs])
    print("%s-%s, %d robots: passes per robot mean %.1f max %d" % (gait, err, n, np.mean([len(s) for s in seqs]), max(len(s) for s in seqs)))
    for name, c in (("slowest robot alone", lone), ("A pass-aligned (kernel)", a), ("B iteration-aligned", b)):
        print("  %-26s mean %.2f  p99 %.2f  max %.2f us above the floor" % (name, c.mean(), np.percentile(c, 99), c.max()))
    worst = np.argsort(-a)[:5]
    for w in worst:
        print("  wavefront %4d: A %.2f B %.2f | %s" % (w, a[w], b[w], seqs[4 * w:4 * w + 4]))


if __name__ == "__main__":
    main()
