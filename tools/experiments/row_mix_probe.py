#!/usr/bin/env python3
"""One wavefront of the balance kernel with different companions for its slowest robot: H = the hard robot, e = an easy
robot (no pass), x = a robot without a stance leg (no QP at all).  Microseconds per launch."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from variant_bench import SLOWEST, timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    ctx = capi.Context(device=0)
    for wl, (hard, easy) in SLOWEST.items():
        gait, err = wl.split("-")
        full = synth.make_states(4096, gait, errors=None if gait == "trot" else err)
        res = []
        for mix in ("HHHH", "HHHe", "HHee", "Heee", "eHee", "eeeH", "Hxxx", "HHxx", "eeee", "HHHHHHHH", "HeeeHeee", "HeeeHHHH"):
            idx = [hard if ch == "H" else easy for ch in mix]
            st = {k: np.ascontiguousarray(v[idx]) for k, v in full.items()}
            for i, ch in enumerate(mix):
                if ch == "x":
                    st["stance"][i] = 0
            d = capi.to_device(st)
            B = len(mix)
            tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
            status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
            us = timed(lambda cap: ctx.balance_solve_device(d, tau, None, status, stream=cap), 200)
            res.append("%s %.2f" % (mix, us))
        print("%-14s %s" % (wl, " | ".join(res)), flush=True)


if __name__ == "__main__":
    main()
