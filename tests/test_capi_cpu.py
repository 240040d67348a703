"""CPU-only: the C-ABI library loads, exports every symbol include/qlamd.h
declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


@pytest.fixture(scope="module")
def capi():
    from quadruped_locomotion_amd import build, capi
    build.build()
    return capi


def test_exports_match_header(capi):
    hdr = open(os.path.join(ROOT, "include", "qlamd.h")).read()
    declared = set(re.findall(r"\b(qlamd_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS)
    L = capi.lib()
    for sym in declared:
        assert getattr(L, sym) is not None


def test_defaults_match_reference_config(capi, oracle):
    p, o = capi.default_params(), oracle.default_params()
    for name, _ in p._fields_:
        assert np.array_equal(np.ctypeslib.as_array(getattr(p, name)) if hasattr(getattr(p, name), "__len__")
                              else getattr(p, name),
                              np.ctypeslib.as_array(getattr(o, name)) if hasattr(getattr(o, name), "__len__")
                              else getattr(o, name)), name
    assert list(p.kp_trans) == [5000, 5000, 10000] and p.friction == 0.6 and p.min_normal_force == 10
    m = capi.default_robot_model()
    assert m.joint_rpy[1][0][0] == 3.1416 and m.joint_xyz[0][3][2] == 0.23  # literal, truncated (Q7)


def test_no_cpu_fallback(capi):
    if has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(capi.QlamdError) as e:
        capi.Context()
    assert e.value.code == capi.ERR_NO_DEVICE
    assert "no CPU fallback" in capi.strerror(capi.ERR_NO_DEVICE)
    h = C.c_void_p()
    assert capi.lib().qlamd_context_create(None, None, 0, C.byref(h)) == capi.ERR_NOT_LOADED
