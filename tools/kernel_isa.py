#!/usr/bin/env python3
"""Device assembly of one translation unit of the library with the flags of quadruped_locomotion_amd/build.py, and what
is worth knowing about each kernel at a glance: registers, scratch bytes, instructions and broadcast-FMAs per basic block.
usage: kernel_isa.py balance_kernel.hip [-D NAME=VALUE ...] [--blocks KERNEL_SUBSTRING] [--out file.s]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_locomotion_amd import build as qbuild  # noqa: E402


def assemble(tu, defines=(), out=None):
    src = os.path.join(ROOT, "quadruped_locomotion_amd", "csrc", tu)
    out = out or "/tmp/%s.s" % os.path.splitext(tu)[0]
    cmd = [qbuild.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.dirname(src)] + ["-D" + d for d in defines] + list(qbuild.TU_FLAGS.get(tu, ())) + [
           "--cuda-device-only", "-S", src, "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


def kernels(path):
    """{kernel name: list of lines}"""
    out, cur, name = {}, None, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
        elif cur is not None:
            cur.append(line.rstrip("\n"))
            if line.startswith(".Lfunc_end"):
                cur = None
    return out


def meta(path):
    txt = open(path).read()
    res = {}
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", txt):
        res[m.group(1)] = {"vgpr": int(m.group(2))}
    for m in re.finditer(r"\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.symbol:\s+(_Z\w+)\.kd", txt):
        res.setdefault(m.group(2), {})["scratch"] = int(m.group(1))
    for m in re.finditer(r"\.agpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.name:\s+(_Z\w+)", txt):
        res.setdefault(m.group(2), {})["agpr"] = int(m.group(1))
    return res


def blocks(lines):
    out, cur = [], None
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = {"label": m.group(1), "n": 0, "fmac_dpp": 0, "lds": 0, "br": []}
            out.append(cur)
            continue
        s = l.split(";")[0].strip()
        if not s or s.startswith(".") or cur is None:
            continue
        cur["n"] += 1
        cur["fmac_dpp"] += "v_fmac_f64_dpp" in s
        cur["lds"] += s.startswith("ds_")
        if s.startswith("s_cbranch") or s.startswith("s_branch"):
            cur["br"].append(s.split()[0][2:] + "->" + s.split()[-1])
    return out


def main():
    args = sys.argv[1:]
    tu = args[0]
    defines = [args[i + 1] for i, a in enumerate(args) if a == "-D"]
    want = [args[i + 1] for i, a in enumerate(args) if a == "--blocks"]
    out = [args[i + 1] for i, a in enumerate(args) if a == "--out"]
    path = assemble(tu, defines, out[0] if out else None)
    md = meta(path)
    ks = kernels(path)
    for name, lines in ks.items():
        if name not in md:
            continue
        n = sum(1 for l in lines if l.split(";")[0].strip() and not l.strip().startswith((".", ";")) and not re.match(r"^\.?\w+:", l))
        print("%-110s vgpr %3d agpr %3d scratch %4d B  %5d instructions" % (name[:110], md[name].get("vgpr", -1), md[name].get("agpr", 0), md[name].get("scratch", 0), n))
        if any(w in name for w in want):
            for b in blocks(lines):
                print("    %-12s n=%4d fmac_dpp=%3d lds=%2d  %s" % (b["label"], b["n"], b["fmac_dpp"], b["lds"], " ".join(b["br"])))
    print(path)


if __name__ == "__main__":
    main()
