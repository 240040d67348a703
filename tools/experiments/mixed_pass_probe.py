#!/usr/bin/env python3
"""GPU experiment: what does a pass cost in which the robots of one wavefront disagree (some add a row, some drop one)?
Times, as single-wavefront launches, groups of 4 consecutive robots of the static-survey bench batch and each of their
robots alone (four copies), with the add / drop sequence of every robot known from the offline restatement
(tools/experiments/active_set_paths.py).  Least squares over the groups then prices the three kinds of lockstep pass.
usage: mixed_pass_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# wavefront -> pass sequences of its four robots ('a' add, 'd' drop), static-survey batch, seed 20261002
WAVES = {
    830: ['aaaaaaadadaadadadadaaadadada', 'aaaaaaaaadada', 'aaaaada', 'aaaaaaaadada'],
    542: ['aaaaaaadaadadadaada', 'aaaaaadaada', 'aaaaaaaaaadada', 'aaaaaaadaddaadaadadaaddaaa'],
    100: ['aaaaaaadaadada', 'aaaaaaaadaa', 'aaaa', 'aaaaaaadadadadda'],
    200: ['aaaaaaa', 'aaaaaaadadadadadadaa', 'aaaaadaaa', 'aaaaaaaada'],
    300: ['', 'aaaaaaadadaadadaaa', 'aaaaaaaadada', 'aaa'],
    400: ['aaaaaaadada', 'aa', 'aaaaaaaddadada', 'aaadaaaa'],
}


def kinds(group):
    n = max(len(s) for s in group)
    aa = dd = mix = 0
    for t in range(n):
        k = {s[t] for s in group if len(s) > t}
        aa += k == {'a'}
        dd += k == {'d'}
        mix += len(k) == 2
    return aa, dd, mix


def main():
    import torch
    from quadruped_locomotion_amd import capi, synth
    from tools.tail_probe import CASES  # noqa: F401  (same timing method)
    ctx = capi.Context(device=0)
    full = synth.make_states(4096, "static", errors="survey")

    def timed(idx, reps=200):
        st = {k: np.ascontiguousarray(v[idx]) for k, v in full.items()}
        d = capi.to_device(st)
        tau = torch.zeros(4, 12, dtype=torch.float64, device="cuda:0")
        status = torch.zeros(4, dtype=torch.int32, device="cuda:0")
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
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        return float(np.median(ts))

    rows, rhs = [], []
    floor = timed([8, 8, 8, 8])          # robot 8 of the batch needs no pass
    print("floor (robot 8, no pass): %.2f us" % floor)
    for w, seqs in WAVES.items():
        t_group = timed([4 * w + k for k in range(4)])
        alone = [timed([4 * w + k] * 4) for k in range(4)]
        aa, dd, mix = kinds(seqs)
        print("wavefront %4d: group %.2f us | alone %s | lockstep passes: %d all-add, %d all-drop, %d mixed | robots %s"
              % (w, t_group, " ".join("%.2f" % a for a in alone), aa, dd, mix, seqs))
        rows.append([aa, dd, mix]); rhs.append(t_group - floor)
        for k in range(4):
            rows.append([seqs[k].count('a'), seqs[k].count('d'), 0]); rhs.append(alone[k] - floor)
    c, res, *_ = np.linalg.lstsq(np.array(rows, float), np.array(rhs), rcond=None)
    print("least squares over %d timings: all-add pass %.2f us, all-drop pass %.2f us, mixed pass %.2f us (rms residual %.2f us)"
          % (len(rhs), c[0], c[1], c[2], float(np.sqrt(np.mean((np.array(rows) @ c - np.array(rhs)) ** 2)))))


if __name__ == "__main__":
    main()
