#!/usr/bin/env python3
"""A 4096-robot batch of easy robots (no pass) with ONE busy wavefront, in different compositions and positions: does the
launch follow that wavefront's own stream?  H = the slowest robot of the calm batch, e = an easy robot."""
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
    for wl in ("static-calm", "trot-survey"):
        hard, easy = SLOWEST[wl]
        gait, err = wl.split("-")
        full = synth.make_states(4096, gait, errors=None if gait == "trot" else err)
        res = []
        for mix, wave in (("eeee", 0), ("Heee", 0), ("HHHH", 0), ("Heee", 230), ("HHHH", 230), ("Heee", 1023), ("HHHH", 1023)):
            idx = np.full(4096, easy)
            for k, ch in enumerate(mix):
                if ch == "H":
                    idx[4 * wave + k] = hard
            st = {k: np.ascontiguousarray(v[idx]) for k, v in full.items()}
            d = capi.to_device(st)
            tau = torch.zeros(4096, 12, dtype=torch.float64, device="cuda:0")
            status = torch.zeros(4096, dtype=torch.int32, device="cuda:0")
            us = timed(lambda cap: ctx.balance_solve_device(d, tau, None, status, stream=cap), 200)
            res.append("%s@%d %.2f" % (mix, wave, us))
        # the real batch with its slowest wavefront replaced by easy robots
        print("%-12s %s" % (wl, " | ".join(res)), flush=True)


if __name__ == "__main__":
    main()
