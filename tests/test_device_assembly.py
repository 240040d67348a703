"""Checks on the device assembly of the library (CPU: hipcc cross-compiles without a GPU).

The lane-cooperative kernels issue `v_fmac_f64_dpp` from inline assembly, where the compiler's hazard recognizer cannot see the
DPP operand: the two wait states between a VALU write of a register and a DPP read of it are the source's own business
(csrc/coop_lanes.hpp, fmac_bc's kNop protocol).  A change of the instruction scheduler or of the loop structure moves
instructions around those statements, so every build is checked over all paths of the control-flow graph."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools import check_dpp_hazards, kernel_isa  # noqa: E402
from quadruped_locomotion_amd import build as qbuild  # noqa: E402

needs_hipcc = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not found")


def test_checker_finds_planted_hazards():
    asm = """
	v_mul_f64 v[2:3], v[0:1], v[0:1]
	s_nop 0
	v_fmac_f64_dpp v[4:5], v[2:3], v[6:7] row_newbcast:0 row_mask:0xf bank_mask:0xf
	s_cbranch_scc0 .LBB0_2
	v_mov_b32_e32 v9, 0
.LBB0_2:
	v_mov_b32_dpp v8, v9 row_ror:8 row_mask:0xf bank_mask:0xf
	v_add_f64 v[2:3], v[0:1], v[0:1]
	s_nop 1
	v_fmac_f64_dpp v[4:5], v[2:3], v[6:7] row_newbcast:0 row_mask:0xf bank_mask:0xf
	v_cmpx_gt_f64 v[0:1], v[2:3]
	s_nop 3
	v_mov_b32_dpp v8, v10 row_ror:8 row_mask:0xf bank_mask:0xf
""".split("\n")
    n, problems = check_dpp_hazards.check_kernel("k", asm)
    assert n == 4
    found = {(p[2].split()[0], p[3].split()[0], p[4]) for p in problems}
    assert ("v_fmac_f64_dpp", "v_mul_f64", 1) in found          # one wait state instead of two
    assert ("v_mov_b32_dpp", "v_mov_b32_e32", 0) in found       # through the fall-through edge only
    assert ("v_mov_b32_dpp", "v_cmpx_gt_f64", 4) in found       # EXEC written by the VALU: five wait states, s_nop 3 gives four
    assert not any(p[2].startswith("v_fmac") and p[3].startswith("v_add_f64") for p in problems)  # s_nop 1 is enough


@needs_hipcc
@pytest.mark.parametrize("tu", qbuild.SOURCE_NAMES)
def test_no_dpp_hazard_in_the_library(tu, tmp_path):
    path = kernel_isa.assemble(tu, out=str(tmp_path / (tu + ".s")))
    total, problems = check_dpp_hazards.check_file(path)
    assert total > 100 or tu != "balance_kernel.hip"
    assert not problems, problems[:5]
    if tu == "balance_kernel.hip":
        # the hot kernel: two wavefronts per SIMD for the latency form (at most 256 registers, nothing spilled to scratch
        # memory), three for the throughput form that large batches take (at most 168 registers; since it carries the QP in
        # two forms -- 12 variables, and 6 for wavefronts whose robots stand on two legs -- the inputs of the second form wait
        # in LDS while the first runs; in the warm-started kernel two or three values go to scratch around it instead, never
        # inside a loop).  The warm-started kernels CALL the cold second attempt of a rejected robot (balance_cold_retry: a
        # function of its own that ends the wavefront); its frame -- the registers the calling convention makes it save --
        # is in the kernel's scratch size, which costs nothing while nobody touches it (profiles/r6/ab_retry_forms.txt); what
        # counts is the scratch traffic of the kernel's own body: none in the latency forms, the six accesses of round 5 in
        # the throughput form.
        md = kernel_isa.meta(path)
        hot = {k: v for k, v in md.items() if "balance_coop_kernel" in k}
        assert len(hot) == 9   # per-leg normals / latency form / throughput form, each plain, placed and placed + warm start
        code = kernel_isa.kernels(path)
        assert sum("balance_cold_retry" in k for k in code) == 3
        for name, m in hot.items():
            three = "ELi3E" in name
            assert m["vgpr"] + m.get("agpr", 0) <= (168 if three else 256), (name, m)
            warm = name.endswith("ELb1ELb1EEEvPKN5qlamd12DeviceParamsENS_9StatePtrsElPdS6_Pi")
            assert m.get("scratch", 0) <= (512 if warm else 0), (name, m)
            body = [l for l in code[name] if "scratch_" in l.split(";")[0]]
            assert len(body) <= ((8 if three else 1) if warm else 0), (name, body)
            assert sum("s_swappc" in l for l in code[name]) == (1 if warm else 0), name
            in_loop = False
            for line in code[name]:
                if line.startswith(".LBB"):
                    in_loop = "in Loop" in line
                assert not (in_loop and "scratch_" in line), (name, line)
