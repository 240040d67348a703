#!/usr/bin/env python3
"""Times every wavefront (4 consecutive robots) of a 4096-robot bench batch on its own: the launch lasts as long as its
slowest wavefront plus the launch ramp, so this is the list a change to the active-set loop has to move.
usage: wave_scan.py [--lib ...] [--presets static-calm,static-survey,trot-survey] [--top 8] [--reps 40]"""
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
    ap.add_argument("--lib", default=None)
    ap.add_argument("--presets", default="static-calm,static-survey,trot-survey")
    ap.add_argument("--top", type=int, default=8)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--waves", type=int, default=1024)
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    ctx = capi.Context(device=0)
    for wl in args.presets.split(","):
        gait, err = wl.split("-")
        full = synth.make_states(4096, gait, errors=None if gait == "trot" else err)
        dfull = capi.to_device(full)
        tau = torch.zeros(4096, 12, dtype=torch.float64, device="cuda:0")
        status = torch.zeros(4096, dtype=torch.int32, device="cuda:0")
        whole = timed(lambda cap: ctx.balance_solve_device(dfull, tau, None, status, stream=cap), 100)
        times = np.zeros(args.waves)
        for w in range(args.waves):
            d = {k: v[4 * w:4 * w + 4] for k, v in dfull.items()}
            t4 = tau[4 * w:4 * w + 4]
            s4 = status[4 * w:4 * w + 4]
            times[w] = timed(lambda cap: ctx.balance_solve_device(d, t4, None, s4, stream=cap), args.reps, replays=3)
        order = np.argsort(-times)[:args.top]
        print("%-14s launch %.2f us | slowest wavefronts: %s | mean %.2f" % (
            wl, whole, "  ".join("%d: %.2f" % (w, times[w]) for w in order), times.mean()), flush=True)


if __name__ == "__main__":
    main()
