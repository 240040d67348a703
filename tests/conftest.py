import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def goldens():
    return np.load(os.path.join(ROOT, "tests", "golden", "qp_goldens.npz"))


class Mirror:
    """Host build of the kernel arithmetic (tests/host_mirror) -- test infrastructure."""

    def __init__(self):
        d = os.path.join(ROOT, "tests", "host_mirror")
        so = os.path.join(d, "libmirror.so")
        srcs = [os.path.join(d, "mirror.cpp"),
                os.path.join(ROOT, "quadruped_locomotion_amd", "csrc", "balance_core.hpp"),
                os.path.join(ROOT, "quadruped_locomotion_amd", "csrc", "params_build.hpp"),
                os.path.join(d, "gi_core.hpp"), os.path.join(d, "gi6_core.hpp"), os.path.join(d, "pose_one_lane.hpp"),
                os.path.join(ROOT, "quadruped_locomotion_amd", "csrc", "pose_core.hpp"),
                os.path.join(ROOT, "quadruped_locomotion_amd", "csrc", "leg_state_core.hpp"),
                os.path.join(ROOT, "quadruped_locomotion_amd", "csrc", "wire_core.hpp"),
                os.path.join(ROOT, "quadruped_locomotion_amd", "csrc", "swing_core.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                                   "-I" + os.path.join(ROOT, "quadruped_locomotion_amd", "csrc"), "-I" + d, "-o", so, srcs[0]])
        self.L = C.CDLL(so)

    def balance(self, O, state, normals=None):
        dp = C.POINTER(C.c_double)
        B = state["q"].shape[0]
        prm = O.default_params()
        arrs = [np.ascontiguousarray(state[n], dtype=np.float64) for n, _ in O.STATE_FIELDS]
        st = np.ascontiguousarray(state["stance"], dtype=np.uint8)
        tau, grf = np.zeros((B, 12)), np.zeros((B, 12))
        status, it, na = (np.zeros(B, np.int32) for _ in range(3))
        nw = None
        if normals is not None:
            nwa = np.ascontiguousarray(normals, dtype=np.float64)
            nw = nwa.ctypes.data_as(dp)
        ip = C.POINTER(C.c_int32)
        self.L.mirror_balance_batch(C.byref(prm), C.c_int64(B), *[a.ctypes.data_as(dp) for a in arrs],
                                    st.ctypes.data_as(C.POINTER(C.c_uint8)), nw, tau.ctypes.data_as(dp),
                                    grf.ctypes.data_as(dp), status.ctypes.data_as(ip), it.ctypes.data_as(ip),
                                    na.ctypes.data_as(ip))
        return tau, grf, status, it, na


@pytest.fixture(scope="session")
def mirror():
    return Mirror()
