#!/usr/bin/env python3
"""The placed entry (qlamd_balance_solve_placed_batch) on the GPU: what the in-kernel indirection costs, what the library's
two placements give, what a grid of other placements would give, what the placement kernel costs, and the loop a caller
would run (solve with the order of the previous step, then the placement for the next one).  Hints are the library's own
`iterations` output of an untimed first solve.
  grid: the hardest p % of the robots one per wavefront (hardest first) with the three easiest each; the rest either in index
        order or sorted (hardest first) behind them.
usage: placed_probe.py [--lib ...] [--cases static-survey:4096,...] [--grid]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from variant_bench import timed  # noqa: E402


def grid_placement(it, p, rest_sorted):
    B = len(it)
    rank = np.argsort(-it, kind="stable")
    nh = min(int(round(p * B)), B // 4)
    hard, easy = rank[:nh], rank[::-1][:3 * nh]
    mid = rank[nh:B - 3 * nh]
    if not rest_sorted:
        mid = np.sort(mid)
    out = np.empty(B, dtype=np.int32)
    out[0:4 * nh:4] = hard
    for k in range(3):
        out[1 + k:4 * nh:4] = easy[k::3]
    out[4 * nh:] = mid
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--cases", default="static-survey:4096,static-calm:4096,trot-survey:4096,trot-survey:8192,trot-survey:16384,trot-survey:65536")
    ap.add_argument("--grid", action="store_true")
    ap.add_argument("--reps", type=int, default=100)
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    ctx = capi.Context(device=0)
    for case in args.cases.split(","):
        wl, n = case.split(":")
        n = int(n)
        gait, err = wl.split("-")
        state = synth.make_states(n, gait, errors=None if gait == "trot" else err)
        d = capi.to_device(state)
        tau = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
        status = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        iters = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        order = torch.arange(n, dtype=torch.int32, device="cuda:0")
        ctx.balance_solve_placed_device(d, tau, None, status, iterations=iters, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        it = iters.cpu().numpy()

        def run(o, with_iters=False):
            if o is not None:
                order.copy_(torch.from_numpy(np.ascontiguousarray(o, dtype=np.int32)))
                torch.cuda.synchronize()
            return timed(lambda cap: ctx.balance_solve_placed_device(d, tau, None, status, order=None if o is None else order,
                                                                      iterations=iters if with_iters else None, stream=cap), args.reps)
        res = ["plain %.2f" % timed(lambda cap: ctx.balance_solve_device(d, tau, None, status, stream=cap), args.reps)]
        res.append("placed/identity %.2f" % run(np.arange(n)))
        res.append("placed/NULL+iterations %.2f" % run(None, True))
        for name, pol in (("latency", capi.PLACEMENT_LATENCY), ("throughput", capi.PLACEMENT_THROUGHPUT)):
            res.append("%s %.2f" % (name, run(ctx.placement_from_iterations(it, policy=pol))))
        # the placement kernel alone, and the loop of a caller: solve (order of the previous step, iterations out) + placement
        pk = timed(lambda cap: ctx.placement_from_iterations(iters, order=order, policy=capi.PLACEMENT_AUTO, stream=cap), args.reps)
        res.append("placement kernel %.2f" % pk)

        def loop(cap):
            ctx.balance_solve_placed_device(d, tau, None, status, order=order, iterations=iters, stream=cap)
            ctx.placement_from_iterations(iters, order=order, policy=capi.PLACEMENT_AUTO, stream=cap)
        res.append("solve+placement per step %.2f" % timed(loop, args.reps))
        # the same loop with the placement made inside the solve's launch (one extra wavefront, in its shadow)
        orders = [order, order.clone()]
        its = [iters, iters.clone()]

        def shadow(cap):
            for k in range(2):
                ctx.balance_solve_placed_device(d, tau, None, status, order=orders[k & 1], iterations=its[k & 1],
                                                prev_iterations=its[(k - 1) & 1], next_order=orders[(k + 1) & 1],
                                                policy=capi.PLACEMENT_AUTO, stream=cap)
        res.append("solve with the next placement in its shadow %.2f" % (timed(shadow, args.reps // 2) / 2))
        print("%-14s %6d robots (iterations mean %.1f max %d) | %s" % (wl, n, it.mean(), it.max(), " | ".join(res)), flush=True)
        if args.grid:
            for rest_sorted in (False, True):
                row = []
                for p in (0.03125, 0.0625, 0.125, 0.1875, 0.25):
                    row.append("%.1f%%: %.2f" % (100 * p, run(grid_placement(it, p, rest_sorted))))
                print("   grid, rest %-6s | %s" % ("sorted" if rest_sorted else "index", " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
