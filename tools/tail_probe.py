#!/usr/bin/env python3
"""Latency of ONE wavefront of the balance kernel as a function of the active-set work of its robots.

The 4096-robot step lasts as long as its slowest wavefront; this probe launches single-wavefront batches (the same
robot four times) for robots of the bench batches with known iteration counts and prints microseconds per launch
(hipGraph of launches; the constant launch gap is in every line, differences are pure kernel time).
usage: tail_probe.py [--lib path/to/libqlamd_variant.so] [--gait static|trot] [--errors calm|survey]
The iteration counts come from tests/golden-free oracle runs done in the build container and are passed in as a
JSON file (tools/tail_probe_cases.json: {"static-calm": [[robot, iters, n_active], ...], ...})."""
import argparse
import json
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
    cases = json.load(open(os.path.join(ROOT, "tools", "tail_probe_cases.json")))
    ctx = capi.Context(device=0)
    for name, rows in cases.items():
        gait, err = name.split("-")
        full = synth.make_states(4096, gait, errors=None if gait == "trot" else err)
        print("== %s (lib %s)" % (name, os.path.basename(capi.LIB_PATH)))
        for robot, iters, nact in rows:
            st = {k: np.ascontiguousarray(np.repeat(v[robot:robot + 1], 4, axis=0)) for k, v in full.items()}
            d = capi.to_device(st)
            tau = torch.zeros(4, 12, dtype=torch.float64, device="cuda:0")
            status = torch.zeros(4, dtype=torch.int32, device="cuda:0")
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    cap = torch.cuda.current_stream().cuda_stream
                    for _ in range(args.reps):
                        ctx.balance_solve_device(d, tau, None, status, stream=cap)
            torch.cuda.current_stream().wait_stream(side)
            g.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / args.reps)
            print("robot %5d  iters %2d  active %2d  status %d   %.2f us" % (robot, iters, nact, int(status[0].item()), float(np.median(ts))))


if __name__ == "__main__":
    main()
