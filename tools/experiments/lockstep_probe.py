#!/usr/bin/env python3
"""GPU experiment: what does sharing a wavefront cost the robots of the bench batch?

The balance kernel puts 4 robots on one wavefront; they pass through the active-set loop in lockstep, so a wavefront
lasts as long as the union of its robots' passes.  This probe times the same 4096 hard robots (a) as they are,
(b) two per wavefront -- the batch interleaved with robots that need no pass at all, [hard, hard, easy, easy] -- and
(c) one per wavefront, [hard, easy, easy, easy].  (b) and (c) put 2 and 4 wavefronts on every SIMD, which is what a
layout with fewer robots per wavefront would do at this batch size.
usage: lockstep_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timed(ctx, capi, torch, state, reps=50):
    B = state["q"].shape[0]
    d = capi.to_device(state)
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            cap = torch.cuda.current_stream().cuda_stream
            for _ in range(reps):
                ctx.balance_solve_device(d, tau, None, status, stream=cap)
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
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    assert (status.cpu().numpy() == 0).all()
    return float(np.median(ts))


def interleave(hard, easy, pattern):
    """pattern: per wavefront slot 'h' or 'e'; consumes the hard robots in order, easy ones cyclically."""
    nh = pattern.count("h")
    groups = hard["q"].shape[0] // nh
    out = {}
    for k in hard:
        rows = []
        for g in range(groups):
            h = 0
            for slot, c in enumerate(pattern):
                if c == "h":
                    rows.append(hard[k][g * nh + h]); h += 1
                else:
                    rows.append(easy[k][(4 * g + slot) % easy[k].shape[0]])
        out[k] = np.ascontiguousarray(np.stack(rows))
    return out


def main():
    import torch
    from quadruped_locomotion_amd import capi, synth
    ctx = capi.Context(device=0)
    calm = synth.make_states(4096, "static", errors="calm")
    # robots of the calm batch whose unconstrained minimiser is feasible: no pass at all
    from oracle import oracle as O
    _, _, st = O.balance_batch(calm)
    it = np.array([O.balance_step(calm, i)["iters"] for i in range(512)])
    easy_idx = np.where(it <= 1)[0]
    easy = {k: v[easy_idx] for k, v in calm.items()}
    print("easy robots: %d of the first 512 calm ones" % len(easy_idx))
    for name, hard in (("static-survey", synth.make_states(4096, "static", errors="survey")), ("trot", synth.make_states(4096, "trot"))):
        # the easy robots take the hard batch's stance pattern? no: they keep their own four-leg stance (cheapest case)
        a = timed(ctx, capi, torch, hard)
        b = timed(ctx, capi, torch, interleave(hard, easy, "hhee"))
        c = timed(ctx, capi, torch, interleave(hard, easy, "heee"))
        e = timed(ctx, capi, torch, {k: np.ascontiguousarray(np.tile(v, (4096 // v.shape[0] + 1, 1))[:4096]) for k, v in easy.items()})
        print("%-14s 4 per wavefront %.2f us | 2 per wavefront (8192 launched) %.2f us | 1 per wavefront (16384 launched) %.2f us | "
              "4096 easy robots %.2f us" % (name, a, b, c, e))


if __name__ == "__main__":
    main()
