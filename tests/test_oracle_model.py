"""Leg FK / Jacobian / gravity restatement (KDL is absent: parity unpinned at that
boundary; these are self-consistency checks) and the kindr helpers."""
import numpy as np
import pytest


def test_fk_nominal_pose(oracle):
    # at q = 0 the chain is T(xyz0) R0 T(0) R1 T(0.308) T(0.308,0,foot_z) with literal rpy
    for leg, (sx, sy, fz) in enumerate([(1, 1, 0.23), (1, -1, 0.22053), (-1, -1, 0.22053), (-1, 1, 0.23)]):
        p, R = oracle.leg_fk(leg, [0.0, 0.0, 0.0])
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
        # Ry(pi/2) maps link x to -z: the straight leg hangs 0.616 m below the hip, offset sideways by foot_z
        assert abs(p[0] - sx * 0.427) < 1e-4 and abs(p[2] - (-0.0095 - 0.616)) < 1e-4
        assert abs(abs(p[1]) - (0.075 + fz)) < 1e-4 and np.sign(p[1]) == sy


@pytest.mark.parametrize("leg", range(4))
def test_jacobian_is_fk_derivative(oracle, leg):
    rng = np.random.default_rng(leg)
    for _ in range(20):
        q = rng.uniform(-1.5, 1.5, 3)
        J = oracle.leg_jacobian(leg, q)
        h = 1e-6
        Jn = np.zeros((3, 3))
        for k in range(3):
            dq = np.zeros(3); dq[k] = h
            Jn[:, k] = (oracle.leg_fk(leg, q + dq)[0] - oracle.leg_fk(leg, q - dq)[0]) / (2 * h)
        assert np.abs(J - Jn).max() < 1e-8


@pytest.mark.parametrize("leg", range(4))
def test_gravity_is_potential_gradient(oracle, leg):
    rng = np.random.default_rng(10 + leg)
    for _ in range(20):
        q = rng.uniform(-1.5, 1.5, 3)
        g = rng.normal(size=3) * 5
        G = oracle.leg_gravity(leg, q, g)
        h = 1e-6
        Gn = np.zeros(3)
        for k in range(3):
            dq = np.zeros(3); dq[k] = h
            Gn[k] = (oracle.leg_potential(leg, q + dq, g) - oracle.leg_potential(leg, q - dq, g)) / (2 * h)
        assert np.abs(G - Gn).max() < 1e-7


def test_quaternion_helpers(oracle):
    rng = np.random.default_rng(3)
    for _ in range(50):
        rv = rng.normal(size=3) * 0.7
        ang = np.linalg.norm(rv)
        q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * rv / ang])
        R = oracle.quat_to_matrix(q)
        K = np.array([[0, -rv[2], rv[1]], [rv[2], 0, -rv[0]], [-rv[1], rv[0], 0]]) / ang
        assert np.allclose(R, np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K, atol=1e-14)
        # boxMinus(exp(rv) * b, b) == rv  (PoseOptimizationSQPTest.cpp:27-37 pins this to 1e-3)
        b = rng.normal(size=4); b /= np.linalg.norm(b)
        a = np.array([q[0] * b[0] - q[1:] @ b[1:], *(q[0] * b[1:] + b[0] * q[1:] + np.cross(q[1:], b[1:]))])
        assert np.allclose(oracle.quat_box_minus(a, b), rv, atol=1e-12)
    assert np.allclose(oracle.quat_box_minus([1, 0, 0, 0], [1, 0, 0, 0]), 0)


def test_virtual_wrench_gravity_only(oracle):
    """Zero tracking error, level base: F_B = (0,0,(27+4*6)*9.8), T_B = 0
    (VirtualModelController.cpp:162-188, quadruped_state.cpp:28,36-41)."""
    z3 = np.zeros((1, 3))
    s = dict(q=np.zeros((1, 12)), base_pos=z3, base_quat=np.array([[1.0, 0, 0, 0]]), base_linvel=z3, base_angvel=z3,
             des_pos=z3, des_quat=np.array([[1.0, 0, 0, 0]]), des_linvel=z3, des_angvel=z3,
             stance=np.ones((1, 4), np.uint8))
    w = oracle.virtual_wrench(s, 0)
    assert np.allclose(w, [0, 0, 51 * 9.8, 0, 0, 0], atol=1e-12)
    # 1 cm too low: both vertical P terms act: 2 * 10000 * 0.01
    s["des_pos"] = np.array([[0, 0, 0.01]])
    assert abs(oracle.virtual_wrench(s, 0)[2] - (51 * 9.8 + 200.0)) < 1e-9
