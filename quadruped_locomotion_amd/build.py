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
SOURCE_NAMES = ("balance_kernel.hip", "pose_kernel.hip", "tick_kernel.hip", "wholebody_kernel.hip")
SOURCES = [os.path.join(_PKG, "csrc", f) for f in SOURCE_NAMES]
OBJ_DIR = os.path.join(_PKG, "csrc", "_obj")
# Code-generation flags per translation unit.  The active-set kernels are single-wavefront latency problems (a
# lone wavefront issues an independent instruction every 5.5 cycles and a dependent one every 8.5), so LLVM's max-ILP
# scheduling strategy, which orders for instruction-level parallelism instead of register pressure, is worth 5-6 % on the
# slowest robot's stream (measured, round 3: 23.2 -> 21.8 us).  Not for pose_kernel.hip (no effect measured).  The
# whole-body solve kernel carries __launch_bounds__(64, 2) so that the strategy cannot push it past 256 registers, where a
# SIMD holds one wavefront instead of two (65 536 robots: 103 -> 141 us without the cap).
TU_FLAGS = {
    "balance_kernel.hip": ("-mllvm", "-amdgpu-sched-strategy=max-ilp"),
    "tick_kernel.hip": ("-mllvm", "-amdgpu-sched-strategy=max-ilp"),
    "wholebody_kernel.hip": ("-mllvm", "-amdgpu-sched-strategy=max-ilp"),
}


def headers():
    csrc = os.path.join(_PKG, "csrc")
    return [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hpp")] + [
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
    return any(os.path.getmtime(f) > t for f in SOURCES + headers() + [os.path.abspath(__file__)])  # the flags live here


def build(force=False, verbose=False, defines=(), lib=None, extra_flags=()):
    """One object per translation unit (compiled side by side), then one link.  `defines` / `lib` build a variant
    (e.g. defines=("QLAMD_STAMPS",) for the diagnostic build, extra_flags=("-mllvm", "...") for a code-generation
    experiment) without touching the product library."""
    out = lib or LIB
    if not (force or lib or needs_build()):
        return out
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = OBJ_DIR if not lib else OBJ_DIR + "_" + os.path.basename(out)
    os.makedirs(obj_dir, exist_ok=True)
    common = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(_ROOT, "include"),
              "-I" + os.path.join(_PKG, "csrc")] + ["-D" + d for d in defines] + list(extra_flags)
    sources = SOURCES  # the diagnostic build (QLAMD_STAMPS) keeps the units and their flags: one stamp buffer per unit
    objs = [os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o") for src in sources]

    def compile_one(pair):
        src, obj = pair
        cmd = common + list(TU_FLAGS.get(os.path.basename(src), ())) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(sources)) as pool:
        list(pool.map(compile_one, zip(sources, objs)))
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
