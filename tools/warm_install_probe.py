#!/usr/bin/env python3
"""What a warm start costs ONE wavefront, by the size of the working set it installs: single-wavefront launches (the same robot
four times) of robots of the static-survey bench batch with 0 .. 12 active rows, warm-started from their own final set (nothing
to do but the installs, one selection that finds nothing, the refinement and the final check) against the cold start of the
same robot -- microseconds per launch in a hipGraph of launches (the launch gap is in every line; differences are kernel time).
usage: warm_install_probe.py [--lib path/to/libqlamd_variant.so] [--reps N]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    ctx = capi.Context(device=0)
    B = 4096
    full = synth.make_states(B, "static", errors="survey")
    d = capi.to_device(full)
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    ws = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    it = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    ctx.balance_solve_placed_device(d, tau, None, status, iterations=it, working_set=ws)
    torch.cuda.synchronize()
    rows = np.array([bin(int(w) & 0xFFFFF).count("1") for w in ws.cpu().numpy().view(np.uint32)])
    iters = it.cpu().numpy()
    print("== static-survey, lib %s: robots by active rows of their final set" % os.path.basename(capi.LIB_PATH))
    print("%5s %6s %6s %10s %10s %12s" % ("rows", "robot", "iters", "cold us", "warm us", "us / install"))
    base = None
    for n in range(0, 13):
        cand = np.nonzero(rows == n)[0]
        if len(cand) == 0:
            continue
        robot = int(cand[np.argsort(iters[cand])[len(cand) // 2]])   # the one with the median cold count
        st = {k: np.ascontiguousarray(np.repeat(v[robot:robot + 1], 4, axis=0)) for k, v in full.items()}
        d1 = capi.to_device(st)
        t1 = torch.zeros(4, 12, dtype=torch.float64, device="cuda:0")
        s1 = torch.zeros(4, dtype=torch.int32, device="cuda:0")
        w1 = ws[robot:robot + 1].repeat(4).contiguous()
        w_out = torch.zeros(4, dtype=torch.int32, device="cuda:0")
        res = []
        for warm in (False, True):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    cap = torch.cuda.current_stream().cuda_stream
                    for _ in range(args.reps):
                        ctx.balance_solve_placed_device(d1, t1, None, s1, prev_working_set=w1 if warm else None, working_set=w_out, stream=cap)
            torch.cuda.current_stream().wait_stream(side)
            g.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / args.reps)
            res.append(float(np.median(ts)))
        if n == 0:
            base = res[1]
        print("%5d %6d %6d %10.2f %10.2f %12s" % (n, robot, iters[robot], res[0], res[1], "%.3f" % ((res[1] - base) / n) if n and base else "-"))


if __name__ == "__main__":
    main()
