#!/usr/bin/env python3
"""What a placement of the robots into wavefronts is worth on the GPU, before anything is built: the bench batches are
PERMUTED ON THE HOST (slot s of the launch holds robot order[s]) and run through the unchanged library, so the figure is the
launch without the cost of an in-kernel indirection.  Hints and placements as in placement_model.py; the hint is the
number of outer iterations of each robot from the numpy restatement of the loop (active_set_paths.py).
Also: every one of the `--lone` hardest robots alone in its wavefront (three companions that need no pass), which is the
floor of any placement.
usage: placement_probe.py [--lib ...] [--cases static-survey:4096,trot-survey:8192,...] [--lone 24]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from variant_bench import timed  # noqa: E402
import placement_model as M  # noqa: E402


def hints_of(state, n):
    seqs = [M.L.sequence(M.P.qp_of(state, i)) if state["stance"][i].any() else "" for i in range(n)]
    return np.array([s.count("a") for s in seqs]), np.array([len(s) for s in seqs])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--cases", default="static-survey:4096,static-calm:4096,trot-survey:4096,trot-survey:8192,trot-survey:16384")
    ap.add_argument("--lone", type=int, default=24)
    ap.add_argument("--reps", type=int, default=100)
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    ctx = capi.Context(device=0)

    def launch_us(state, reps):
        d = capi.to_device(state)
        B = state["q"].shape[0]
        tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
        status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        return timed(lambda cap: ctx.balance_solve_device(d, tau, None, status, stream=cap), reps)

    for case in args.cases.split(","):
        wl, n = case.split(":")
        n = int(n)
        gait, err = wl.split("-")
        state = synth.make_states(n, gait, errors=None if gait == "trot" else err)
        adds, passes = hints_of(state, n)
        res = []
        kinds = ["identity", "top5", "top12.5", "top25", "snake", "sorted", "sorted_easy_first"]
        for kind in kinds:
            if kind == "sorted_easy_first":
                slots = M.place("sorted", adds, n)[::-1].copy()
            else:
                slots = M.place(kind, adds, n)
            st = {k: np.ascontiguousarray(v[slots]) for k, v in state.items()}
            res.append("%s %.2f" % (kind, launch_us(st, args.reps)))
        print("%-14s %6d robots (adds mean %.1f max %d) | %s" % (wl, n, adds.mean(), adds.max(), " | ".join(res)), flush=True)
        if n == 4096 and args.lone:
            easy = int(np.argmin(passes))
            hard = np.argsort(-passes, kind="stable")[:args.lone]
            out = []
            for h in hard:
                idx = [int(h), easy, easy, easy]
                st = {k: np.ascontiguousarray(v[idx]) for k, v in state.items()}
                out.append("%d (%d passes): %.2f" % (h, passes[h], launch_us(st, 200)))
            print("%-14s hardest robots alone in a wavefront: %s" % (wl, "  ".join(out)), flush=True)


if __name__ == "__main__":
    main()
