#!/usr/bin/env python3
"""Builds variants of the library into scratch_bin/ (git-ignored, travels to the GPU box) for A/B timing with
tools/experiments/variant_bench.py:
  variants.py name=DEFINE1,DEFINE2=VALUE ...      e.g.  lone0=QLAMD_LONE_FORM=0  acc2=QLAMD_DIR_ACC=2
  variants.py --rev HEAD name                     the sources of a git revision (csrc + include), current build flags"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadruped_locomotion_amd import build as qbuild  # noqa: E402


def main():
    os.makedirs(os.path.join(ROOT, "scratch_bin"), exist_ok=True)
    args = sys.argv[1:]
    if args and args[0] == "--rev":
        rev, name = args[1], args[2]
        tmp = tempfile.mkdtemp()
        subprocess.check_call("git -C %s archive %s quadruped_locomotion_amd/csrc include | tar -x -C %s" % (ROOT, rev, tmp), shell=True)
        out = os.path.join(ROOT, "scratch_bin", "libqlamd_%s.so" % name)
        objs = []
        for tu in qbuild.SOURCE_NAMES:
            obj = os.path.join(tmp, tu + ".o")
            subprocess.check_call([qbuild.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(tmp, "include"),
                                   "-I" + os.path.join(tmp, "quadruped_locomotion_amd", "csrc")] + list(qbuild.TU_FLAGS.get(tu, ())) +
                                  ["-c", os.path.join(tmp, "quadruped_locomotion_amd", "csrc", tu), "-o", obj], stderr=subprocess.DEVNULL)
            objs.append(obj)
        subprocess.check_call([qbuild.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, stderr=subprocess.DEVNULL)
        shutil.rmtree(tmp)
        print(out)
        return
    for spec in args:
        name, _, defs = spec.partition("=")
        out = os.path.join(ROOT, "scratch_bin", "libqlamd_%s.so" % name)
        qbuild.build(force=True, defines=tuple(d for d in defs.split(",") if d), lib=out)
        print(out)


if __name__ == "__main__":
    main()
