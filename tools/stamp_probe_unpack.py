"""Diagnostic: s_memtime stamps (shader clock, 2.41 GHz) at the segment boundaries of robot_state_unpack_kernel, block 0 lane 0
(needs variants/libqlamd_stamps.so built with -DQLAMD_STAMPS)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from quadruped_locomotion_amd import capi
# diagnostic build: python -c "from quadruped_locomotion_amd import build; build.build(defines=('QLAMD_STAMPS',), lib='variants/libqlamd_stamps.so')"
capi.LIB_PATH = os.path.join(ROOT, "variants", "libqlamd_stamps.so")
from test_wire_format import random_message, batch_of_messages
B = 4096
one, _ = random_message(np.random.default_rng(3), ragged=True)
names = ["offsets->stage+sync", "template check", "walk/extract", "template out+sync", "write-out"]
for label, (blob, off) in (("uniform", (one * B, np.arange(B + 1, dtype=np.int64) * len(one))), ("ragged", batch_of_messages(B, 1)[:2])):
    ctx = capi.Context()
    for rep in range(3):
        capi.robot_state_unpack(ctx, blob, off)
        out = (C.c_ulonglong * 32)()
        capi.lib().qlamd_debug_stamps_tick(out, 32)
        t = np.array(out[20:26], dtype=np.float64)
        print(label, "launch", rep, " ".join("%s %.2f us;" % (names[k], (t[k + 1] - t[k]) / 2408.0) for k in range(5)), "total %.2f us" % ((t[5] - t[0]) / 2408.0))
