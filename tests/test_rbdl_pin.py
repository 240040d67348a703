"""The recursive Newton-Euler pass (SURVEY.md rows a18 / f1, and the recursion shared with the whole-body entries) pinned on
the reference's own RBDL output: tests/golden/rbdl_leg_id.npz holds the 10 001 (q, qd, qdd) rows of
single_leg_test/DataFloder/PlannedData.txt, the torques RBDL's InverseDynamics wrote for them into
TauofInversedynamics.txt (MyRobotSolver::IDynamicsCalculation, single_leg_test/lib/model_test_header.cpp:277-301) and the
3-link model of model_initialization (:183-222); made by tests/tools/gen_rbdl_leg_id.py.

The file prints six significant digits, so "reproduces" means a relative deviation of at most half a unit of the sixth
digit: 5e-6."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRINT_PRECISION = 5.0e-6
_dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def rbdl():
    return np.load(os.path.join(ROOT, "tests", "golden", "rbdl_leg_id.npz"))


def _p(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def chain_of(d, split_last_body):
    """The five per-segment arrays of one leg.  The reference model has three bodies; the fourth segment (the fixed end
    link of a qlamd_robot_model leg) is massless as stored, or -- for arithmetic that divides by a link's mass -- carries
    half of body c at the same centre of mass, which is the same rigid body."""
    xyz, rpy, mass, com, inertia = (d[k].copy() for k in ("joint_xyz", "joint_rpy", "link_mass", "link_com", "link_inertia"))
    if split_last_body:
        mass[3] = mass[2] = 0.5 * mass[2]
        com[3] = com[2]
        inertia[3] = inertia[2] = 0.5 * inertia[2]
    return xyz, rpy, mass, com, inertia


def relative_deviation(tau, ref):
    return float((np.abs(tau - ref) / np.abs(ref)).max())


def oracle_rnea_rows(O, d, chain):
    m = [_p(a) for a in chain]
    g = _p(d["gravity"])
    out = np.zeros((d["q"].shape[0], 3))
    a = [_p(d[k]) for k in ("q", "qd", "qdd")]
    O.lib().oracle_chain_rnea_rows(C.c_long(out.shape[0]), *[x[1] for x in m], a[0][1], a[1][1], a[2][1], g[1],
                                   out.ctypes.data_as(_dp))
    return out


def test_fixture_is_the_reference_run(rbdl):
    assert rbdl["q"].shape == rbdl["qd"].shape == rbdl["qdd"].shape == rbdl["tau"].shape == (10001, 3)
    # first row as printed by the reference (TauofInversedynamics.txt line 1)
    assert np.array_equal(rbdl["tau"][0], [-2.59659, -2.67911, -0.468734])
    assert np.array_equal(rbdl["q"][0], [-1.2, -1.0, -0.4])
    assert np.abs(rbdl["tau"]).min() > 1e-4            # no entry near zero: a relative bound is meaningful everywhere


def test_oracle_rnea_reproduces_all_rbdl_torques(oracle, rbdl):
    tau = oracle_rnea_rows(oracle, rbdl, chain_of(rbdl, False))
    assert relative_deviation(tau, rbdl["tau"]) <= PRINT_PRECISION
    # the split representation of body c is the same rigid body
    tau2 = oracle_rnea_rows(oracle, rbdl, chain_of(rbdl, True))
    assert np.abs(tau2 - tau).max() < 1e-13


class _SP(C.Structure):
    _fields_ = [("kp", C.c_double * 3), ("kd", C.c_double * 3), ("period", C.c_double), ("accel_window", C.c_double),
                ("accel_scale", C.c_double), ("gravity", C.c_double)]


def robot_model_of(chain, cls):
    """qlamd_robot_model with the chain on all four legs."""
    xyz, rpy, mass, com, inertia = chain
    m = cls()
    for leg in range(4):
        for k in range(4):
            m.link_mass[leg][k] = mass[k]
            for c in range(3):
                m.joint_xyz[leg][k][c] = xyz[k][c]
                m.joint_rpy[leg][k][c] = rpy[k][c]
                m.link_com[leg][k][c] = com[k][c]
            for c in range(6):
                m.link_inertia[leg][k][c] = inertia[k][c]
    m.base_mass = 1.0
    return m


def test_product_arithmetic_on_host_reproduces_rbdl_torques(mirror, rbdl):
    """csrc/swing_core.hpp (host build) through the swing-leg entry's own formula: gains 0, and the oldest queue entry
    chosen so that its finite difference is the file's acceleration (model_test_header.cpp:417-431,460)."""
    from quadruped_locomotion_amd import capi
    model = robot_model_of(chain_of(rbdl, True), capi.RobotModel)
    sp = _SP((0.0,) * 3, (0.0,) * 3, 1.0, 1.0, 1.0, 9.81)
    rows = np.arange(0, 10001, 7)
    worst = 0.0
    zero = np.zeros(3)
    for i in rows:
        q, qd = rbdl["q"][i].copy(), rbdl["qd"][i].copy()
        qd_old = qd - rbdl["qdd"][i]
        tau = np.zeros(3)
        mirror.L.mirror_swing_leg_model(C.byref(model), 0, C.byref(sp), _p(q)[1], _p(q)[1], _p(qd)[1], _p(qd_old)[1],
                                        _p(zero)[1], _p(zero)[1], tau.ctypes.data_as(_dp))
        worst = max(worst, relative_deviation(tau, rbdl["tau"][i]))
    assert worst <= PRINT_PRECISION * 1.01      # qd - (qd - qdd) differs from qdd in the last bit


@pytest.mark.gpu
def test_swing_kernel_reproduces_all_rbdl_torques(rbdl):
    """qlamd_swing_leg_torque_batch on the GPU with the reference's RBDL test model passed through qlamd_robot_model."""
    from quadruped_locomotion_amd import capi
    B = rbdl["q"].shape[0]
    ctx = capi.Context(model=robot_model_of(chain_of(rbdl, True), capi.RobotModel))
    prm = capi.default_swing_params()
    for k in range(3):
        prm.kp[k] = 0.0
        prm.kd[k] = 0.0
    prm.period, prm.accel_window, prm.accel_scale, prm.gravity = 1.0, 1.0, 1.0, 9.81
    q = np.tile(rbdl["q"], (1, 4))
    qd = np.tile(rbdl["qd"], (1, 4))
    qd_old = np.tile(rbdl["qd"] - rbdl["qdd"], (1, 4))
    zero = np.zeros((B, 12))
    support = np.zeros((B, 4), dtype=np.uint8)          # every leg swings: four copies of the chain per row
    tau = capi.swing_leg_torque(ctx, q, qd, qd_old, zero, zero, support, params=prm)
    for leg in range(4):
        assert relative_deviation(tau[:, 3 * leg:3 * leg + 3], rbdl["tau"]) <= PRINT_PRECISION * 1.01
    # the reference's production setting of the same entry: half the finite-difference acceleration (:460)
    prm.accel_scale = 0.5
    qd_old2 = np.tile(rbdl["qd"] - 2.0 * rbdl["qdd"], (1, 4))
    tau2 = capi.swing_leg_torque(ctx, q, qd, qd_old2, zero, zero, support, params=prm)
    assert relative_deviation(tau2[:, 0:3], rbdl["tau"]) <= PRINT_PRECISION * 1.01
    ctx.close()


def test_wholebody_recursion_agrees_with_the_pinned_chain_recursion(oracle):
    """oracle_wholebody.c's inverse dynamics is a different formulation (spatial vectors in link coordinates, floating
    base).  With the base held at rest its joint rows are four fixed-base chains -- exactly what oracle_chain_rnea,
    pinned on RBDL above, computes -- velocity-product and inertia terms included."""
    rng = np.random.default_rng(7)
    ident = np.array([1.0, 0.0, 0.0, 0.0])
    for _ in range(20):
        q, qd, qdd = rng.uniform(-1.2, 1.2, 12), rng.uniform(-3, 3, 12), rng.uniform(-20, 20, 12)
        nu, nudot = np.concatenate([np.zeros(6), qd]), np.concatenate([np.zeros(6), qdd])
        full = oracle.wb_inverse_dynamics(q, ident, nu, nudot, gravity=9.81)
        for leg in range(4):
            sl = slice(3 * leg, 3 * leg + 3)
            chain = oracle.leg_rnea(leg, q[sl], qd[sl], qdd[sl], np.array([0.0, 0.0, -9.81]))
            assert np.abs(full[6 + 3 * leg:9 + 3 * leg] - chain).max() < 1e-11
