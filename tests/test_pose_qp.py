"""PoseOptimizationQP and PoseConstraintsChecker: the oracle against the reference's own
known-answer tests (free_gait_core/test/PoseOptimizationQpTest.cpp), the kernel arithmetic compiled for the
host against the oracle, and (gpu) the device entries against the oracle."""
import ctypes as C

import numpy as np
import pytest

from quadruped_locomotion_amd import synth
from test_pose_sqp_host import _PoseParamsDev

HIPS, ORDER = synth.POSE_HIPS, synth.POSE_LEG_ORDER
# limb ids 0..3 = LF, RF, RH, LH.  The reference tests name LH = (-1,-0.5) and RH = (-1,0.5).
NOMINAL = np.array([[1.0, 0.5, 0.0], [1.0, -0.5, 0.0], [-1.0, 0.5, 0.0], [-1.0, -0.5, 0.0]])


def problem(stance, nominal, pose0, polygon=None, nv=None):
    stance = np.asarray(stance, dtype=float)
    poly = np.zeros((4, 2))
    if polygon is None:
        # checkSupportRegion (PoseOptimizationBase.cpp:52-58): vertices from the support stance in its map order
        polygon, nv = stance[list(ORDER), :2], 4
    poly[:len(polygon)] = polygon
    return dict(stance=stance[None], stance_mask=np.ones((1, 4), np.uint8), nominal=np.asarray(nominal, float)[None],
                polygon=poly[None].copy(), n_vertices=np.array([nv], np.int32), r_com=np.zeros((1, 3)),
                max_len=np.full((1, 4), 10.0), pose=np.asarray(pose0, float)[None])


def yaw(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


def known_answer_cases():
    cases = []
    # quadrupedSymmetricUnconstrained (:20-52): expect (0, 0, 0.3)
    nominal = NOMINAL + [0, 0, -0.4]
    cases.append(("symmetric", problem(NOMINAL + [0, 0, -0.1], nominal, [0, 0, 0, 1, 0, 0, 0]), [0.0, 0.0, 0.3]))
    # quadrupedSymmetricWithOffsetUnconstrained (:54-85): start pose is the optimum
    cases.append(("offset", problem(NOMINAL + [30.0, 20.0, 10.0], NOMINAL, [30, 20, 10, 1, 0, 0, 0]), [30.0, 20.0, 10.0]))
    # quadrupedSymmetricWithYawUnconstrained (:122-150)
    q = [np.cos(0.25), 0, 0, np.sin(0.25)]
    cases.append(("yaw", problem(NOMINAL @ yaw(0.5).T, NOMINAL, [0, 0, 0] + q), [0.0, 0.0, 0.0]))
    return cases


@pytest.mark.parametrize("name,pb,expect", known_answer_cases(), ids=[c[0] for c in known_answer_cases()])
def test_oracle_known_answers(oracle, name, pb, expect):
    r = oracle.pose_qp(pb, 0, HIPS, ORDER)
    assert r["status"] == 0
    assert np.allclose(r["pose"][:3], expect, atol=1e-3)          # the reference's own tolerance
    assert np.array_equal(r["pose"][3:], pb["pose"][0, 3:])       # orientation untouched


def constrained_problem():
    # PoseOptimizationQpTest.cpp:184-217: roll 0.5, LF moved out, support triangle LH, RH, RF
    Rx = np.array([[1, 0, 0], [0, np.cos(0.5), -np.sin(0.5)], [0, np.sin(0.5), np.cos(0.5)]])
    feet = np.array([[2.0, 0.5, 0.0], [1.0, -0.5, 0.0], [-1.0, -0.5, 0.0], [-1.0, 0.5, 0.0]]) @ Rx.T
    tri = feet[[3, 2, 1], :2]
    return problem(feet, NOMINAL, [0, 0, 0, np.cos(0.25), np.sin(0.25), 0, 0], polygon=tri, nv=3), tri


def test_oracle_constrained_inside_region(oracle):
    pb, tri = constrained_problem()
    r = oracle.pose_qp(pb, 0, HIPS, ORDER)
    assert r["status"] == 0
    c = tri.mean(0)
    grown = c + (tri - c) * (1 + 1e-4)                            # offsetInward(-1e-5) in the reference test
    assert oracle.lib().oracle_polygon_is_inside(3, grown.ctypes.data_as(C.POINTER(C.c_double)),
                                                 (C.c_double * 2)(*r["pose"][:2])) == 1
    # the unconstrained optimum (mean residual) lies outside, so a constraint is active
    free = (pb["stance"][0] - NOMINAL @ np.array([[1, 0, 0], [0, np.cos(0.5), -np.sin(0.5)],
                                                  [0, np.sin(0.5), np.cos(0.5)]]).T).mean(0)
    assert np.abs(free[:2] - r["pose"][:2]).max() > 1e-3


def mirror_aux(mirror, mode, pb, min_len=None, leg_tol=0.0, sfo=None, full=False):
    B = pb["pose"].shape[0]
    P = _PoseParamsDev()
    for l in range(4):
        for a in range(3):
            P.hips[l][a] = HIPS[l][a]
        P.leg_order[l] = ORDER[l]
    P.com_weight, P.tol, P.max_iter, P.dummy_equality = 2.0, 0.05, 30, 1
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    out = np.zeros((B, 7)); st = np.zeros(B, np.int32); ok = np.zeros(B, np.uint8)
    stage = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
    mn = None if min_len is None else np.ascontiguousarray(min_len, dtype=np.float64)
    sf = None if sfo is None else np.ascontiguousarray(sfo, dtype=np.float64)
    mirror.L.mirror_pose_aux_batch(
        C.c_int(mode), C.byref(P), C.c_int64(B), pb["stance"].ctypes.data_as(dp),
        pb["stance_mask"].ctypes.data_as(C.POINTER(C.c_uint8)), pb["nominal"].ctypes.data_as(dp), pb["polygon"].ctypes.data_as(dp),
        pb["n_vertices"].ctypes.data_as(ip), pb["r_com"].ctypes.data_as(dp), pb["max_len"].ctypes.data_as(dp),
        pb["pose"].ctypes.data_as(dp), mn.ctypes.data_as(dp) if mn is not None else None, C.c_double(leg_tol),
        out.ctypes.data_as(dp), st.ctypes.data_as(ip), ok.ctypes.data_as(C.POINTER(C.c_uint8)),
        sf.ctypes.data_as(dp) if sf is not None else None, stage.ctypes.data_as(ip), it.ctypes.data_as(ip))
    if full:
        return out, st, stage, it
    return out, st, ok


def check_inputs(pb, seed=5):
    """Poses to check: the start poses jittered so that both verdicts occur; per-leg minimum lengths."""
    rng = np.random.default_rng(seed)
    B = pb["pose"].shape[0]
    poses = pb["pose"].copy()
    poses[:, :2] += rng.normal(scale=0.08, size=(B, 2))
    poses[:, 2] += rng.normal(scale=0.05, size=B)
    min_len = rng.uniform(0.15, 0.32, size=(B, 4))
    return dict(pb, pose=poses), min_len


def test_kernel_math_on_host_matches_oracle(oracle, mirror):
    pb = synth.make_pose_problems(256)
    out, st, _ = mirror_aux(mirror, 1, pb)
    for i in range(256):
        r = oracle.pose_qp(pb, i, HIPS, ORDER)
        assert r["status"] == st[i] == 0
        assert np.abs(r["pose"] - out[i]).max() < 1e-12
    chk, min_len = check_inputs(pb)
    _, _, ok = mirror_aux(mirror, 2, chk, min_len, 0.01)
    want = np.array([oracle.pose_check(chk, i, chk["pose"][i], HIPS, ORDER, min_len[i], 0.01) for i in range(256)])
    assert np.array_equal(ok, want)
    assert 0 < want.sum() < 256


def test_known_answers_through_kernel_math(oracle, mirror):
    for name, pb, expect in known_answer_cases():
        out, st, _ = mirror_aux(mirror, 1, pb)
        assert st[0] == 0 and np.allclose(out[0, :3], expect, atol=1e-3), name
    pb, tri = constrained_problem()
    out, st, _ = mirror_aux(mirror, 1, pb)
    assert st[0] == 0 and np.abs(out[0] - oracle.pose_qp(pb, 0, HIPS, ORDER)["pose"]).max() < 1e-12


@pytest.mark.gpu
def test_device_pose_qp_and_check_match_oracle(oracle):
    from quadruped_locomotion_amd import capi
    ctx = capi.Context()
    B = 1000                                                       # ragged last wave (16 problems per wave)
    pb = synth.make_pose_problems(B)
    pose, st = capi.pose_qp(ctx, pb)
    assert (st == 0).all()
    for i in range(B):
        r = oracle.pose_qp(pb, i, HIPS, ORDER)
        assert r["status"] == 0 and np.abs(r["pose"] - pose[i]).max() < 1e-9
    chk, min_len = check_inputs(pb)
    ok = capi.pose_check(ctx, chk, min_len, 0.01)
    want = np.array([oracle.pose_check(chk, i, chk["pose"][i], HIPS, ORDER, min_len[i], 0.01) for i in range(B)])
    assert np.array_equal(ok, want) and 0 < want.sum() < B
    # min_limb_length = NULL means 0
    ok0 = capi.pose_check(ctx, chk, None, 0.01)
    want0 = np.array([oracle.pose_check(chk, i, chk["pose"][i], HIPS, ORDER, (0, 0, 0, 0), 0.01) for i in range(B)])
    assert np.array_equal(ok0, want0)
    for name, kp, expect in known_answer_cases():
        p1, s1 = capi.pose_qp(ctx, kp)
        assert s1[0] == 0 and np.allclose(p1[0, :3], expect, atol=1e-3), name
    cp, tri = constrained_problem()
    p1, s1 = capi.pose_qp(ctx, cp)
    assert s1[0] == 0 and np.abs(p1[0] - oracle.pose_qp(cp, 0, HIPS, ORDER)["pose"]).max() < 1e-9
    # empty batch is a no-op
    empty = {k: v[:0] for k, v in pb.items()}
    p0, s0 = capi.pose_qp(ctx, empty)
    assert p0.shape == (0, 7)
