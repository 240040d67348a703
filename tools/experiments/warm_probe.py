#!/usr/bin/env python3
"""The warm-started loop of a caller on the GPU: every step ONE qlamd_balance_solve_placed_batch launch that starts each robot's
QP from its final working set of the previous step, runs in the placement made from the previous steps' (residual) iteration
counts and leaves both for the next step.  Odd steps run on the states one control period (2.5 ms) later than even steps, so
that every hint comes from OTHER states, as at 400 Hz.  Microseconds per step (captured graph) next to the plain entry and
the placed loop without warm start.  usage: warm_probe.py [--cases static-survey:4096,...]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from variant_bench import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="static-survey:4096,static-calm:4096,trot-survey:4096,trot-survey:8192,trot-survey:16384,trot-survey:65536")
    ap.add_argument("--reps", type=int, default=100)
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    ctx = capi.Context(device=0)
    for case in args.cases.split(","):
        wl, n = case.split(":")
        n = int(n)
        gait, err = wl.split("-")
        sA = synth.make_states(n, gait, errors=None if gait == "trot" else err)
        ds = [capi.to_device(sA), capi.to_device(synth.next_tick_states(sA, 0.0025))]
        tau = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
        status = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        i32 = lambda: torch.zeros(n, dtype=torch.int32, device="cuda:0")  # noqa: E731
        orders = [torch.arange(n, dtype=torch.int32, device="cuda:0") for _ in range(2)]
        its, wss = [i32(), i32()], [i32(), i32()]

        def loop(warm, placed):
            def f(cap):
                for k in range(2):
                    ctx.balance_solve_placed_device(
                        ds[k], tau, None, status, stream=cap,
                        order=orders[k & 1] if placed else None, iterations=its[k & 1],
                        prev_iterations=its[(k - 1) & 1] if placed else None, next_order=orders[(k + 1) & 1] if placed else None,
                        policy=capi.PLACEMENT_AUTO,
                        prev_working_set=wss[(k - 1) & 1] if warm else None, working_set=wss[k & 1] if warm else None)
            return f
        res = ["plain %.2f" % timed(lambda cap: ctx.balance_solve_device(ds[0], tau, None, status, stream=cap), args.reps)]
        for name, warm, placed in (("placed", False, True), ("warm", True, False), ("warm + placed (hint = installs + passes)", True, True)):
            for w in wss + its:
                w.zero_()
            us = timed(loop(warm, placed), args.reps // 2) / 2
            torch.cuda.synchronize()
            it = its[1].cpu().numpy()
            res.append("%s %.2f (iterations mean %.2f max %d, status ok %s)" % (name, us, it.mean(), it.max(), bool((status == 0).all().item())))
        print("%-14s %6d robots | %s" % (wl, n, " | ".join(res)), flush=True)


if __name__ == "__main__":
    main()
