#!/usr/bin/env python3
"""Would two robots per wavefront (two wavefronts per SIMD at 4096 robots) beat four?  Emulated without a new kernel: the
4096 robots of a preset spread over 8192 slots, two real robots and two easy ones (no pass: they finish after the first
selection and ride along as ghost rows) per wavefront, against the 4096-robot launch.  Also one real robot per wavefront."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from variant_bench import SLOWEST, timed  # noqa: E402


def main():
    import torch
    from quadruped_locomotion_amd import capi, synth
    ctx = capi.Context(device=0)

    def run(st):
        B = st["q"].shape[0]
        d = capi.to_device(st)
        tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
        status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        return timed(lambda cap: ctx.balance_solve_device(d, tau, None, status, stream=cap), 100)

    for wl in ("static-survey", "trot-survey", "static-calm"):
        hard, easy = SLOWEST[wl]
        gait, err = wl.split("-")
        full = synth.make_states(4096, gait, errors=None if gait == "trot" else err)
        pick = lambda idx: {k: np.ascontiguousarray(v[idx]) for k, v in full.items()}  # noqa: E731
        four = run(full)
        idx2 = np.full(8192, easy)
        idx2.reshape(2048, 4)[:, :2] = np.arange(4096).reshape(2048, 2)
        two = run(pick(idx2))
        idx1 = np.full(16384, easy)
        idx1.reshape(4096, 4)[:, 0] = np.arange(4096)
        one = run(pick(idx1))
        print("%-14s four robots per wavefront %6.2f us | two (+ two easy) %6.2f | one (+ three easy) %6.2f" % (wl, four, two, one), flush=True)


if __name__ == "__main__":
    main()
