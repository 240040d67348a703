"""The "support legs first" bookkeeping of csrc/balance_coop.hpp on the CPU: the order of the legs in a robot's row for every
support mask (support legs ascending, then the others), the table the device reads it from, and working sets carried to slot
order and back -- the host halves of the header's functions, built with hipcc (no GPU needed), against a Python restatement."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def slot_legs(mask):
    legs = [l for l in range(4) if (mask >> l) & 1] + [l for l in range(4) if not (mask >> l) & 1]
    return legs


def to_slots(ws, legs, kinds):
    rows = (1 << kinds) - 1
    out = 0
    for sl, leg in enumerate(legs):
        out |= ((ws >> (kinds * leg)) & rows) << (kinds * sl)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_slot_order_and_working_set_maps(tmp_path):
    exe = str(tmp_path / "slot_legs_check")
    src = os.path.join(ROOT, "tests", "cpp", "slot_legs_check.hip")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "quadruped_locomotion_amd", "csrc"), src, "-o", exe],
                          stderr=subprocess.DEVNULL)
    lines = subprocess.check_output([exe], text=True).strip().splitlines()
    assert len(lines) == 16
    for m, line in enumerate(lines):
        v = [int(x) for x in line.split()]
        legs = slot_legs(m)
        assert v[0] == m and v[1:5] == legs and v[5] == 1, line          # the order, and the device's table agrees with it
        ws5 = (0x9A3C5 ^ (m * 0x11111)) & 0xFFFFF
        ws11 = (0x5A5A5A5A5A5 ^ (m * 0x123456789)) & 0xFFFFFFFFFFF
        assert v[6] == to_slots(ws5, legs, 5) and v[7] == to_slots(ws11, legs, 11), line
        assert v[8] == 1 and v[9] == 1, line                              # ... and back again
    assert slot_legs(0b1111) == [0, 1, 2, 3] and slot_legs(0b0101) == [0, 2, 1, 3] and slot_legs(0b1010) == [1, 3, 0, 2]
