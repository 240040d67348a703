#!/usr/bin/env python3
"""Two wavefronts per SIMD (8192 robots): does it matter WHICH two?  If workgroup b of a launch lands on SIMD b mod 1024, the
latency placement (wavefront r carries the r-th hardest robot) makes SIMD-mates of ranks r and r + 1024; pairing rank r with
rank 2047 - r instead evens out the sum per SIMD.  Orders are built on the host from the library's own iteration counts and
handed to the placed entry; us per launch (hipGraph of 20 launches, best of 5 x 10 replays).
usage: simd_mates_probe.py [--lib LIB]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    ctx = capi.Context(device=0)
    for gait, errors, B in (("trot", None, 8192), ("static", "survey", 8192), ("trot", None, 6144), ("trot", None, 12288)):
        state = synth.make_states(B, gait, errors=errors)
        d = capi.to_device(state)
        tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
        status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        iters = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        ctx.balance_solve_placed_device(d, tau, None, status, iterations=iters, stream=stream)
        torch.cuda.synchronize()
        it = iters.cpu().numpy()
        rank_to_robot = np.argsort(-np.clip(it, 0, 23), kind="stable")
        W = (B + 3) // 4

        def build(wave_of_rank):
            order = np.full(4 * W, -1, dtype=np.int64)
            for r in range(W):
                order[4 * wave_of_rank(r)] = rank_to_robot[r]
            # the three easiest left go to the wavefronts in the order of their hard robots
            for r in range(W, B):
                e = B - 1 - r
                order[4 * wave_of_rank(e // 3) + 1 + e % 3] = rank_to_robot[r]
            return order[:B] if (order[:B] >= 0).all() and len(set(order[:B])) == B else None

        S = 1024
        variants = {
            "library (wave r = rank r)": lambda r: r,
            "mates (second 1024 wavefronts reversed)": lambda r: r if r < S or r >= 2 * S else S + (2 * S - 1 - r) if W >= 2 * S else r,
            "mates for W waves (rank r with rank W-1-r)": lambda r: r if r < S else (S + (W - 1 - r)) if W <= 2 * S else r,
            "interleaved (hard ranks on even waves)": lambda r: (2 * r if 2 * r < W else 2 * (r - (W + 1) // 2) + 1),
        }
        out = []
        for name, f in variants.items():
            order = build(f)
            if order is None:
                out.append("%s: n/a" % name)
                continue
            o = torch.from_numpy(order.astype(np.int32)).to("cuda:0")
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    for _ in range(20):
                        ctx.balance_solve_placed_device(d, tau, None, status, order=o, iterations=iters,
                                                        stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            best = 1e9
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                e0.record()
                for _ in range(10):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
            out.append("%s: %.2f" % (name, best))
        print("%s%s %d robots | %s" % (gait, "-" + errors if errors else "", B, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
