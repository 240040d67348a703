#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps (shader clock, 2.41 GHz: tools/ubench/issue_model.hip calibrates it) of pose_sqp_coop_kernel, block 0 lane 0: the prologue and the segments of the
LAST SQP iteration of problem 0 (the bench batch, 5 iterations).  Needs the diagnostic build:
  python -c "from quadruped_locomotion_amd import build; build.build(defines=('QLAMD_STAMPS',), lib='variants/libqlamd_stamps.so')"
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_locomotion_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "variants", "libqlamd_stamps.so")
MHZ = 2408.0
SEG = [(16, 17, "load record (16 lanes) + read back"), (17, 18, "polygon centroid / half-spaces, per-problem sums"),
       (20, 21, "objective: R, sums, gradient, Hessian blocks"), (21, 22, "my row of G, g0; constraint normal and bound"),
       (22, 23, "QP: Gauss-Jordan"), (23, 24, "QP: x0, first selection"), (24, 25, "QP: active-set passes"),
       (25, 26, "QP: refinement"), (26, 27, "(return)"), (27, 28, "pose update (plus), |dp| test"), (29, 30, "store")]


def main():
    ctx = capi.Context()
    pb = synth.make_pose_problems(4096)
    params = capi.default_pose_params()
    params.tolerance, params.max_iterations = 0.0, 5
    first = {k: (v[:4] if hasattr(v, "shape") and v.shape and v.shape[0] == 4096 else v) for k, v in pb.items()}
    for name, batch in (("one wavefront (4 problems)", first), ("4096 problems", pb)):
        for _ in range(3):
            pose, it, st = capi.pose_sqp(ctx, batch, params)
        out = (C.c_ulonglong * 32)()
        capi.lib().qlamd_debug_stamps_pose(out, 32)
        t = np.array(out[:32], dtype=np.float64)
        print("== %s: iterations of problem 0: %d; whole kernel %.0f cycles = %.2f us (stamp 16 -> 30)" % (
            name, int(it[0]), t[30] - t[16], (t[30] - t[16]) / MHZ))
        for a, b, what in SEG:
            print("   %-58s %7.0f cycles %6.2f us" % (what, t[b] - t[a], (t[b] - t[a]) / MHZ))
        print("   %-58s %7.0f cycles %6.2f us" % ("one SQP iteration (20 -> 28)", t[28] - t[20], (t[28] - t[20]) / MHZ))
    print("(arithmetic is free to move across a stamp: the two linearisation segments are small because the compiler sinks\n"
          " most of that work to where the Gauss-Jordan elimination first needs it -- read them together with it)")


if __name__ == "__main__":
    main()
