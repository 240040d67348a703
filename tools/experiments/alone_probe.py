#!/usr/bin/env python3
"""What do a slow robot's three neighbours in its wavefront cost it?  The placed + warm-started loop runs a trajectory to its last
tick (diagnostic build with workgroup stamps, variants/libqlamd_blockstamps.so); the last tick is then run twice from the same
working sets: in the loop's own placement, and with every robot of the K slowest wavefronts ALONE in a wavefront (the other three
rows empty).  Prints, per slow wavefront, its duration against the longest of its four robots alone.
usage: alone_probe.py [--gait static|trot] [--batch 4096] [--ticks 16] [-k 24]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gait", default="static")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--ticks", type=int, default=16)
    ap.add_argument("-k", type=int, default=24)
    ap.add_argument("--lib", default=os.path.join(ROOT, "variants", "libqlamd_blockstamps.so"))
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    capi.LIB_PATH = os.path.abspath(args.lib)
    B, K = args.batch, args.k
    states = synth.trajectory(B, args.gait, args.ticks, errors="survey" if args.gait == "static" else None)
    ctx = capi.Context(device=0)
    L = capi.lib()
    L.qlamd_debug_block_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
    order = [torch.arange(B, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    iters = [torch.zeros(B, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    ws = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    shadows = (B + 1023) // 1024 if B < 16384 else (B + 4095) // 4096
    nw = (B + 3) // 4

    def read(slot, n):
        out = (C.c_ulonglong * n)()
        assert L.qlamd_debug_block_stamps(out, slot, n) == 0
        return np.array(out[:], dtype=np.float64) * 0.01

    for k, s in enumerate(states[:-1]):
        ctx.balance_solve_placed_device(capi.to_device(s), tau, None, status, order=order[k & 1], iterations=iters[k & 1],
                                        prev_iterations=iters[(k - 1) & 1], next_order=order[(k + 1) & 1], policy=capi.PLACEMENT_LATENCY,
                                        prev_working_set=ws, working_set=ws, stream=stream)
    torch.cuda.synchronize()
    k = len(states) - 1
    d = capi.to_device(states[k])
    ws0 = ws.clone()
    ordk = order[k & 1].cpu().numpy().copy()
    # (a) the loop's own placement
    ws_out = torch.zeros_like(ws)
    for rep in range(3):
        ctx.balance_solve_placed_device(d, tau, None, status, order=order[k & 1], iterations=iters[k & 1], prev_iterations=iters[(k - 1) & 1],
                                        next_order=order[(k + 1) & 1], policy=capi.PLACEMENT_LATENCY, prev_working_set=ws0, working_set=ws_out,
                                        stream=stream)
        torch.cuda.synchronize()
    n = min(2048, shadows + nw)
    t0, t3 = read(0, n)[shadows:], read(3, n)[shadows:]
    dur = t3 - t0
    it = iters[k & 1].cpu().numpy()
    slow = np.argsort(-dur)[:K]
    launch_a = t3.max() - t0.min()
    # (b) their robots alone
    robots = [int(r) for w in slow for r in ordk[4 * w:4 * w + 4] if 0 <= r < B]
    alone = np.full(B, -1, np.int32)
    alone[0:4 * len(robots):4] = robots
    rest = [r for r in ordk if r not in set(robots)]
    room = B - 4 * len(robots)
    alone[4 * len(robots):] = rest[:room]
    o2 = torch.from_numpy(alone).to("cuda:0")
    it2 = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    for rep in range(3):
        ctx.balance_solve_placed_device(d, tau, None, status, order=o2, iterations=it2, policy=capi.PLACEMENT_AUTO, prev_working_set=ws0,
                                        working_set=ws_out, stream=stream)
        torch.cuda.synchronize()
    n2 = min(2048, nw)
    u0, u3 = read(0, n2), read(3, n2)
    dur2 = (u3 - u0)[:len(robots)]
    launch_b = u3.max() - u0.min()
    # (c) the same robots alone, started cold (the reference's method from the empty set): what a race of the two starts would give
    for rep in range(3):
        ctx.balance_solve_placed_device(d, tau, None, status, order=o2, iterations=it2, policy=capi.PLACEMENT_AUTO, stream=stream)
        torch.cuda.synchronize()
    c0, c3 = read(0, n2), read(3, n2)
    dur3 = (c3 - c0)[:len(robots)]
    best = np.minimum(dur2, dur3)
    print("alone, the loop's start: slowest %.2f us | alone, cold: slowest %.2f | the better start of each robot: slowest %.2f (the loop's start wins for %d of %d)"
          % (dur2.max(), dur3.max(), best.max(), int((dur2 <= dur3).sum()), len(robots)))
    print("%s, %d robots, last of %d ticks: launch %.2f us in the loop's placement, %.2f with the %d robots of the %d slowest wavefronts alone (and %d robots left out)"
          % (args.gait, B, len(states), launch_a, launch_b, len(robots), K, len(rest) - room))
    print("wavefront: duration | its robots: iterations -> duration alone")
    j = 0
    gains = []
    for w in slow:
        rob = [int(r) for r in ordk[4 * w:4 * w + 4] if 0 <= r < B]
        al = dur2[j:j + len(rob)]
        j += len(rob)
        gains.append(dur[w] - al.max())
        print("  #%-4d %6.2f | %s | longest alone %.2f (%+.2f)" % (w, dur[w], "  ".join("%d: i%d -> %.2f" % (r, it[r], a) for r, a in zip(rob, al)), al.max(), al.max() - dur[w]))
    print("a slow wavefront is %.2f us (median; mean %.2f) longer than its slowest robot alone; the slowest wavefront %.2f, the slowest robot alone %.2f"
          % (np.median(gains), np.mean(gains), dur.max(), dur2.max()))


if __name__ == "__main__":
    main()
