#!/usr/bin/env python3
"""Gather-friendly inputs for permuted launches (round-5 review, item 7): the placed + warm-started loop on a trajectory with the
nine double fields of a robot inside ONE record of R doubles (a library built with -DQLAMD_RECORD_DOUBLES=R: every field pointer
points into the same [B][R] buffer at its field's offset) against the per-field arrays of the shipped library (the layout of
hardware_interface::RobotStateHandle::Data, robot_state_interface.hpp:28-65).  Prints us per step; run under rocprofv3 --pmc
FETCH_SIZE / WRITE_SIZE for the bytes (the kernel's name carries no hint of the layout: one library per process).
usage: packed_record_probe.py LIB R|0 [--gait trot] [--batch 4096] [--steps 100] [--no-graph]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OFFSETS = {"q": 0, "base_pos": 12, "base_quat": 16, "base_linvel": 20, "base_angvel": 23, "des_pos": 26, "des_quat": 30,
           "des_linvel": 34, "des_angvel": 37}   # doubles; quaternions on 16-byte boundaries


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("record", type=int)
    ap.add_argument("--gait", default="static")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    capi.LIB_PATH = os.path.abspath(args.lib)
    B, K, R = args.batch, args.steps, args.record
    T = min(K, 64 if B > 16384 else 200)
    states = synth.trajectory(B, args.gait, T, errors="survey" if args.gait == "static" else None)
    ds = []
    for s in states:
        if R:
            rec = np.zeros((B, R))
            for k, o in OFFSETS.items():
                rec[:, o:o + s[k].shape[1]] = s[k]
            t = torch.from_numpy(rec).to("cuda:0")
            d = {k: t[:, o:] for k, o in OFFSETS.items()}      # views: data_ptr() = the field's first element of robot 0
            d["stance"] = torch.from_numpy(np.ascontiguousarray(s["stance"])).to("cuda:0")
            d["_keep"] = t
        else:
            d = capi.to_device(s)
        ds.append(d)
    ctx = capi.Context(device=0)
    order = [torch.arange(B, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    iters = [torch.zeros(B, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    ws = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    status = torch.zeros(B, dtype=torch.int32, device="cuda:0")

    def step(k, st):
        ctx.balance_solve_placed_device(ds[k % T], tau, None, status, order=order[k & 1], iterations=iters[k & 1],
                                        prev_iterations=iters[(k - 1) & 1], next_order=order[(k + 1) & 1], policy=capi.PLACEMENT_AUTO,
                                        prev_working_set=ws, working_set=ws, stream=st)
    stream = torch.cuda.current_stream().cuda_stream
    for k in range(10):
        step(k, stream)
    torch.cuda.synchronize()
    assert (status == 0).all()
    if args.no_graph:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(K):
            step(k, stream)
        e1.record()
        torch.cuda.synchronize()
        print("%s record %d %s B=%d eager: %.2f us per step" % (os.path.basename(args.lib), R, args.gait, B, e0.elapsed_time(e1) * 1e3 / K))
        return
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            cap = torch.cuda.current_stream().cuda_stream
            for k in range(K):
                step(k, cap)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / K)
    print("%s record %d %s B=%d: %.2f us per step (all status ok: %s)" % (os.path.basename(args.lib), R, args.gait, B, float(np.median(ts)), bool((status == 0).all().item())))


if __name__ == "__main__":
    main()
