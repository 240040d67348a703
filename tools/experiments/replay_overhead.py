#!/usr/bin/env python3
"""What a timed region of K = 20 captured steps (the driver's arguments) pays beyond its kernels: wall time of
synchronize / replay / synchronize under the runtime's default wait, under hipDeviceScheduleSpin, and with the host polling
an event before the synchronize.  usage: replay_overhead.py [--spin]   (--spin must be a process of its own: the flag is
set before the device is initialised)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    spin = "--spin" in sys.argv
    import torch
    if spin:
        hip = C.CDLL("libamdhip64.so")
        rc = hip.hipSetDeviceFlags(C.c_uint(1))  # hipDeviceScheduleSpin
        print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", rc)
    from quadruped_locomotion_amd import capi, synth
    ctx = capi.Context(device=0)
    B = 4096
    d = capi.to_device(synth.make_states(B, "static", errors="survey"))
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    for K in (20, 200):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                cap = torch.cuda.current_stream().cuda_stream
                for _ in range(K):
                    ctx.balance_solve_device(d, tau, None, status, stream=cap)
        torch.cuda.current_stream().wait_stream(side)
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            g.replay()
        torch.cuda.synchronize()
        res = {}
        for mode in ("synchronize", "poll an event, then synchronize"):
            el = []
            for _ in range(21):
                torch.cuda.synchronize()
                ev = torch.cuda.Event()
                t0 = time.perf_counter()
                g.replay()
                if mode != "synchronize":
                    ev.record()
                    while not ev.query():
                        pass
                torch.cuda.synchronize()
                el.append(time.perf_counter() - t0)
            res[mode] = np.median(el) * 1e6 / K
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print("K = %3d %s | events %.2f us per step | %s" % (K, "spin flag" if spin else "default  ", e0.elapsed_time(e1) * 1e3 / K,
                                                           " | ".join("%s %.2f" % kv for kv in res.items())), flush=True)


if __name__ == "__main__":
    main()
