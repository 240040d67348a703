#!/usr/bin/env python3
"""Hazard check of the device assembly: a DPP instruction that reads a VGPR written by a VALU instruction needs two wait
states in between (five after a VALU write of EXEC).  hipcc inserts them for the DPP moves it generates itself, but it
cannot see inside `asm volatile("v_fmac_f64_dpp ...")` (csrc/coop_lanes.hpp: fmac_bc and its kNop protocol), so every
change of the instruction scheduler or of the loop structure is checked here, over all paths of the control-flow graph.
usage: check_dpp_hazards.py [file.s ...]     (no argument: assembles the four translation units with the build's flags)
exit code 1 if a hazard is found."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DPP_RE = re.compile(r"\b(row_newbcast|row_ror|row_shl|row_shr|row_bcast|row_mirror|row_half_mirror|quad_perm|wave_shl|wave_shr|wave_rol|wave_ror)\b")
REG_RE = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")
NEED_VGPR, NEED_EXEC = 2, 5


def regs(op):
    m = REG_RE.match(op.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def parse(lines):
    """-> list of blocks {label, insts: [(text, states, valu_writes, writes_exec, dpp_reads)], succ labels, falls}"""
    blocks, cur = [], {"label": "<entry>", "insts": [], "succ": [], "falls": True}
    blocks.append(cur)
    for raw in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", raw)
        if m:
            cur = {"label": m.group(1), "insts": [], "succ": [], "falls": True}
            blocks.append(cur)
            continue
        s = raw.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        states = 1
        if op == "s_nop":
            states = int(ops[0], 0) + 1
        writes, wexec, dpp = set(), False, set()
        if op.startswith("v_") and op != "v_nop":
            if op.startswith("v_cmpx"):
                wexec = True
            elif ops and not op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
                writes = regs(ops[0])
            if op.startswith(("v_permlane", "v_swap")) and len(ops) > 1:
                writes |= regs(ops[1])
        if DPP_RE.search(s) and op.startswith("v_"):
            # the DPP operand is src0: the first source (the second operand of the instruction)
            src = ops[1].split()[0] if len(ops) > 1 else ""
            dpp = regs(src)
        cur["insts"].append((s, states, writes, wexec, dpp))
        if op.startswith("s_cbranch"):
            cur["succ"].append(ops[-1])
        elif op == "s_branch":
            cur["succ"].append(ops[-1])
            cur["falls"] = False
        elif op in ("s_endpgm", "s_setpc_b64"):
            cur["falls"] = False
    return blocks


def check_kernel(name, lines):
    blocks = parse(lines)
    index = {b["label"]: i for i, b in enumerate(blocks)}
    preds = {i: [] for i in range(len(blocks))}
    for i, b in enumerate(blocks):
        for t in b["succ"]:
            if t in index:
                preds[index[t]].append(i)
        if b["falls"] and i + 1 < len(blocks):
            preds[i + 1].append(i)
    # a conditional branch in the middle of a block also leaves it: the tail seen by the successor is the instructions up to
    # the branch; taking the whole block is conservative only in the number of wait states counted, so cut at the branch
    problems = []

    def tails(bi, upto, need, depth=0):
        """yield lists of instructions (nearest first) reaching back `need` wait states before position `upto` of block bi"""
        b = blocks[bi]
        got, seq = 0, []
        for k in range(upto - 1, -1, -1):
            inst = b["insts"][k]
            seq.append(inst)
            got += inst[1]
            if got >= need:
                yield seq
                return
        if depth > 6 or not preds[bi]:
            yield seq
            return
        for p in preds[bi]:
            pb = blocks[p]
            # position in the predecessor from which control leaves to bi
            cut = len(pb["insts"])
            if not (pb["falls"] and p + 1 == bi):
                for k, inst in enumerate(pb["insts"]):
                    if inst[0].startswith(("s_cbranch", "s_branch")) and inst[0].split()[-1] == b["label"]:
                        cut = k + 1
            for more in tails(p, cut, need - got, depth + 1):
                yield seq + more

    for bi, b in enumerate(blocks):
        for k, inst in enumerate(b["insts"]):
            if not inst[4]:
                continue
            for seq in tails(bi, k, NEED_EXEC):
                dist = 0
                for prev in seq:
                    if dist < NEED_VGPR and (prev[2] & inst[4]):
                        problems.append((name, b["label"], inst[0], prev[0], dist))
                    if dist < NEED_EXEC and prev[3]:
                        problems.append((name, b["label"], inst[0], prev[0], dist))
                    dist += prev[1]
                    if dist >= NEED_EXEC:
                        break
    return len([1 for b in blocks for i in b["insts"] if i[4]]), problems


def check_file(path):
    from tools.kernel_isa import kernels
    total, problems = 0, []
    for name, lines in kernels(path).items():
        n, p = check_kernel(name, lines)
        total += n
        problems += p
    return total, problems


def main():
    paths = sys.argv[1:]
    if not paths:
        from tools.kernel_isa import assemble
        from quadruped_locomotion_amd import build as qbuild
        paths = [assemble(tu) for tu in qbuild.SOURCE_NAMES]
    bad = 0
    for path in paths:
        total, problems = check_file(path)
        seen = set()
        for name, label, inst, prev, dist in problems:
            key = (name, label, inst, prev)
            if key in seen:
                continue
            seen.add(key)
            print("HAZARD %s %s: `%s` %d wait state(s) after `%s`" % (name[:60], label, inst, dist, prev))
        print("%s: %d DPP instructions, %d hazards" % (os.path.basename(path), total, len(seen)))
        bad += len(seen)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
