#!/usr/bin/env python3
"""Which workgroup does the whole tick wait for?  The diagnostic build with workgroup stamps only (-DQLAMD_BLOCK_STAMPS: the
kernels at nearly the shipped pace; -DQLAMD_STAMPS adds the segment stamps inside the solver, which make it 2.7 x slower) stamps
the start and the end of every workgroup of the tick's two launches (robot_state_unpack_kernel with the leg state machine at its
tail; tick_solve_kernel = balance blocks + swing-branch blocks) and the phase boundaries of the parser's blocks with s_memrealtime,
the device-wide 100 MHz counter.  Prints, per launch: when its workgroups start and end relative to the first start of the tick,
the slowest workgroups, what the end of the launch is made of, and the phases of a parser block.  Needs
  python -c "from quadruped_locomotion_amd import build; build.build(defines=('QLAMD_BLOCK_STAMPS',), lib='variants/libqlamd_blockstamps.so')"
usage: stamp_probe_tick_blocks.py [--ragged] [--batch 4096]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--ticks", type=int, default=12)
    ap.add_argument("--lib", default=os.path.join(ROOT, "variants", "libqlamd_blockstamps.so"))
    args = ap.parse_args()
    import torch
    from quadruped_locomotion_amd import capi, synth
    capi.LIB_PATH = os.path.abspath(args.lib)
    B = args.batch
    rng = np.random.default_rng(11)
    blob, off, _ = synth.make_messages(B, ragged=args.ragged)
    s = synth.make_states(B, "trot")
    host = dict(messages=blob, offsets=off, joint_position=s["q"],
                joint_velocity=rng.normal(scale=0.3, size=(B, 12)), joint_velocity_oldest=rng.normal(scale=0.3, size=(B, 12)),
                base_position=s["base_pos"], base_orientation=s["base_quat"], base_linear_velocity=np.ascontiguousarray(s["base_linvel"]),
                base_angular_velocity=np.ascontiguousarray(s["base_angvel"]), contact=rng.integers(0, 2, (B, 4)).astype(np.uint8),
                limb_state=np.zeros((B, 4), np.int8), store_flag=np.zeros((B, 4), np.uint8), stored_joint_position=np.zeros((B, 12)),
                leg_mode=np.zeros((B, 4), np.uint8), support=np.ones((B, 4), np.uint8), pid_error_last=np.zeros((B, 12)),
                pid_error_integral=np.zeros((B, 12)), joint_effort=np.zeros((B, 12)), leg_state_code=np.zeros((B, 4), np.int8),
                status=np.full(B, -1, np.int32), message_status=np.full(B, -1, np.int32), command=np.zeros(capi.tick_command_bytes(B), np.uint8))
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to("cuda:0") for k, v in host.items()}
    ctx = capi.Context(device=0)
    stream = torch.cuda.current_stream().cuda_stream
    L = capi.lib()
    L.qlamd_debug_block_stamps_tick.argtypes = [C.c_void_p, C.c_int, C.c_int]
    n_unpack, n_bal, n_swing = (B + 3) // 4, (B + 3) // 4, (4 * B + 63) // 64

    def read(slot, n):
        out = (C.c_ulonglong * n)()
        assert L.qlamd_debug_block_stamps_tick(out, slot, n) == 0
        return np.array(out[:], dtype=np.float64) * 0.01  # 100 MHz -> us

    summary = []
    for tick in range(args.ticks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        capi.full_tick(ctx, dev, 0.0025, memory=capi.MEM_DEVICE, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        us, ue = read(0, n_unpack), read(1, n_unpack)
        ss, se = read(2, n_bal + n_swing), read(3, n_bal + n_swing)
        t0 = us.min()
        if tick == args.ticks - 1:
            # the unpack blocks' phases (slots 4-7: template checked against memory | staged, if a message missed | walked, stored | written out)
            prev, names = us, ["offsets, template -> fields from memory", "staging in LDS (a miss in the block)", "status, walk of a miss, stores",
                               "write-out of rows parsed by one lane", "state machine, end"]
            print("phases of the unpack blocks of the last tick (us): p50 | p99")
            for nm, t in zip(names, [read(4, n_unpack), read(5, n_unpack), read(6, n_unpack), read(7, n_unpack), ue]):
                print("   %-40s %6.2f %6.2f" % (nm, np.median(t - prev), np.percentile(t - prev, 99)))
                prev = t
        summary.append((e0.elapsed_time(e1) * 1e3, ue.max() - t0, ss.min() - t0, se.max() - t0, se[:n_bal].max() - t0, se[n_bal:].max() - t0))
    print("%d robots, %s, %d ticks; times in us from the first workgroup start of the tick (s_memrealtime, 10 ns steps); HIP events around the tick: median %.1f us"
          % (B, "a layout per message" if args.ragged else "one layout", args.ticks, np.median([x[0] for x in summary[2:]])))
    print("per tick (from the third on), medians: unpack launch ends %.2f | solve launch starts %.2f | ends %.2f (balance blocks %.2f, swing blocks %.2f)"
          % tuple(np.median([x[k] for x in summary[2:]]) for k in range(1, 6)))
    # the last tick in detail
    def describe(name, st, en, t0, groups):
        d = en - st
        print("%s: %d workgroups | start: first %.2f last %.2f | end: p50 %.2f p90 %.2f p99 %.2f max %.2f | duration: p50 %.2f p99 %.2f max %.2f"
              % (name, len(st), st.min() - t0, st.max() - t0, *(np.percentile(en - t0, [50, 90, 99])), en.max() - t0,
                 *np.percentile(d, [50, 99]), d.max()))
        for gname, lo, hi in groups:
            e = en[lo:hi] - t0
            worst = lo + np.argsort(-en[lo:hi])[:5]
            print("   %-16s ends: p50 %.2f p99 %.2f max %.2f | last five: %s" % (
                gname, np.percentile(e, 50), np.percentile(e, 99), e.max(),
                "  ".join("#%d start %.2f end %.2f" % (w, st[w] - t0, en[w] - t0) for w in worst)))
    describe("robot_state_unpack_kernel (+ leg state machine)", us, ue, t0, [("all", 0, n_unpack), ("block 0 (logger)", 0, 1)])
    describe("tick_solve_kernel", ss, se, t0, [("balance blocks", 0, n_bal), ("swing blocks", n_bal, n_bal + n_swing)])
    late = np.sort(ue - t0)[::-1]
    print("unpack launch: %d workgroups end within 1 us of the last one, %d within 2 us; the gap to the first start of the next launch is %.2f us"
          % (int((late > late[0] - 1.0).sum()), int((late > late[0] - 2.0).sum()), ss.min() - ue.max()))


if __name__ == "__main__":
    main()
