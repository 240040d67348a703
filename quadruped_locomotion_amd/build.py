"""Build the HIP extension in-tree (gfx950 only): quadruped_locomotion_amd/libqlamd.so.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container; the built .so travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB = os.path.join(_PKG, "libqlamd.so")
SOURCES = [os.path.join(_PKG, "csrc", f) for f in ("balance_kernel.hip",)]
HEADERS = [os.path.join(_PKG, "csrc", f) for f in ("balance_core.hpp", "balance_coop.hpp", "pose_coop.hpp", "qp_coop.hpp", "params_build.hpp", "gi_core.hpp",
                                                     "gi6_core.hpp", "pose_core.hpp", "swing_core.hpp", "leg_state_core.hpp", "wire_core.hpp")] + [
    os.path.join(_ROOT, "include", f) for f in ("qlamd.h", "qlamd_robot_constants.h")]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIP extension cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not (force or needs_build()):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_PKG, "csrc"),
           "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
