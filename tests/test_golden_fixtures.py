"""Fixtures generated in the build container (tests/tools/gen_*.py; only the .npz travel) against the oracle (CPU) and,
under -m gpu, DIRECTLY against the HIP kernels through the C-ABI:
  sqp_goldens.npz     the pose SQP with every inner QP solved by the reference's own compiled QuadProg++ (dummy equality
                      column on), per-iteration steps: sequencequadraticproblemsolver.cpp:18-102, QuadProg++.cc:52-446
  force_qp_exact.npz  40-digit minimisers of the golden force-distribution QPs: the stand-in for the OOQP boundary
                      (ContactForceDistribution.cpp:490) that cannot be pinned
  model_checks.npz    leg FK / Jacobian / gravity torque from an independent numpy evaluation of the reference URDF
                      (quadrupedkinematics.cpp:143-278,485-552 go through KDL, absent here)
"""
import os

import numpy as np
import pytest

from conftest import ROOT
from quadruped_locomotion_amd import synth

GOLD = os.path.join(ROOT, "tests", "golden")
# Contact forces in newtons against the exact minimiser.  The joint torques follow through J' (entries below 0.7 m), so
# this keeps them an order of magnitude inside the 1e-6 of BASELINE.json's north_star; the reference's own compiled
# QuadProg++ sits 3e-9 from the exact point on these problems (tests/tools/gen_force_qp_exact.py).
FORCE_TOL = 1e-7
HIPS, ORDER = synth.POSE_HIPS, synth.POSE_LEG_ORDER


@pytest.fixture(scope="module")
def sqp():
    return np.load(os.path.join(GOLD, "sqp_goldens.npz"))


@pytest.fixture(scope="module")
def exact():
    return np.load(os.path.join(GOLD, "force_qp_exact.npz"))


@pytest.fixture(scope="module")
def model():
    return np.load(os.path.join(GOLD, "model_checks.npz"))


def sqp_problems(sqp):
    return {k[3:]: sqp[k] for k in sqp.files if k.startswith("pb_")}


def golden_states():
    """The states tests/tools/gen_goldens.py assembled the golden force QPs from, in its order."""
    trot, static = synth.make_states(4096, "trot"), synth.make_states(512, "static")
    four = [("static", i) for i in range(64)] + [("trot", i) for i in range(4096) if trot["stance"][i].sum() == 4][:64]
    two = [("trot", i) for i in range(4096) if trot["stance"][i].sum() == 2][:128]

    def gather(items):
        return {k: np.ascontiguousarray(np.stack([(static if src == "static" else trot)[k][i] for src, i in items])) for k in trot}
    return gather(four), gather(two)


# ------------------------------------------------------------------------------------------------ CPU: the oracle
def test_oracle_sqp_reproduces_the_reference_solver_iteration_by_iteration(oracle, sqp):
    pb = sqp_problems(sqp)
    n = pb["pose"].shape[0]
    for c in range(n):
        r = oracle.pose_sqp(pb, c, HIPS, ORDER)
        k = int(sqp["iters"][c])
        assert r["status"] == 0 and r["iters"] == k
        assert np.array_equal(r["dp"], sqp["dp"][c, :k])            # the restated solver is bit-identical to the compiled one
        assert np.array_equal(r["pose"], sqp["final_pose"][c])
        for it in range(k):                                           # and so is every inner QP on its own
            m = int(sqp["m"][c, it])
            q = oracle.solve_quadprog(sqp["G"][c, it], sqp["g0"][c, it], np.zeros((6, 1)), np.zeros(1), sqp["CI"][c, it][:, :m],
                                      sqp["ci0"][c, it][:m])
            assert q["status"] == 0 and np.array_equal(q["x"], sqp["dp"][c, it])


def test_oracle_force_qp_against_exact_minimisers(oracle, goldens, exact):
    """The pinned Goldfarb-Idnani restatement lands within 5e-9 N of the 40-digit minimiser (its own stopping rule,
    QuadProg++.cc:246, allows a summed violation of ~2e-8), with the same active rows."""
    for name in ("n12", "n6"):
        for b in range(goldens[name + "_g0"].shape[0]):
            r = oracle.solve_quadprog(goldens[name + "_G"][b], goldens[name + "_g0"][b], None, None, goldens[name + "_CI"][b],
                                      goldens[name + "_ci0"][b])
            assert r["status"] == 0
            assert np.abs(r["x"] - exact[name + "_x"][b]).max() < 5e-9
        assert exact[name + "_kkt_residual"].max() < 1e-40


def test_oracle_model_against_independent_urdf_evaluation(oracle, model):
    for n in range(model["q"].shape[0]):
        for l in range(4):
            ql = model["q"][n, 3 * l:3 * l + 3]
            assert np.abs(oracle.leg_fk(l, ql)[0] - model["foot"][n, l]).max() < 1e-14
            assert np.abs(oracle.leg_jacobian(l, ql) - model["jacobian"][n, l]).max() < 1e-14
            assert np.abs(oracle.leg_gravity(l, ql, model["gravity_in_base"][n]) - model["gravity_torque"][n, l]).max() < 1e-13


# ------------------------------------------------------------------------------------------------ GPU: the kernels
@pytest.fixture(scope="module")
def gpu():
    import torch
    from quadruped_locomotion_amd import capi
    assert torch.cuda.is_available(), "these tests need the MI355X"
    ctx = capi.Context(device=0)
    yield capi, ctx, torch
    ctx.close()


@pytest.mark.gpu
def test_device_pose_sqp_and_inner_qps_against_reference_goldens(gpu, sqp):
    capi, ctx, torch = gpu
    pb = sqp_problems(sqp)
    pose, it, st = capi.pose_sqp(ctx, pb)
    assert (st == 0).all() and np.array_equal(it, sqp["iters"])
    assert np.abs(pose - sqp["final_pose"]).max() < 1e-9
    # every inner QP through qlamd_qp_solve_batch with the reference's all-zero equality column
    for m in np.unique(sqp["m"][sqp["m"] > 0]):
        idx = np.argwhere(sqp["m"] == m)
        G = np.stack([sqp["G"][c, k] for c, k in idx]); g0 = np.stack([sqp["g0"][c, k] for c, k in idx])
        CI = np.stack([sqp["CI"][c, k][:, :m] for c, k in idx]); ci0 = np.stack([sqp["ci0"][c, k][:m] for c, k in idx])
        dp = np.stack([sqp["dp"][c, k] for c, k in idx])
        B = len(idx)
        x, f, s = capi.qp_solve(ctx, G, g0, np.zeros((B, 6, 1)), np.zeros((B, 1)), CI, ci0)
        assert (s == 0).all() and np.abs(x - dp).max() < 1e-9
        assert np.allclose(f, np.stack([sqp["f"][c, k] for c, k in idx]), rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_device_force_qp_against_exact_minimisers(gpu, goldens, exact):
    """Both device routes to the force QP against the 40-digit minimisers: the dense QP batch on the golden matrices and
    the balance kernel itself (contact forces of the states the goldens were assembled from)."""
    capi, ctx, torch = gpu
    for name in ("n12", "n6"):
        x, f, st = capi.qp_solve(ctx, goldens[name + "_G"], goldens[name + "_g0"], None, None, goldens[name + "_CI"], goldens[name + "_ci0"])
        err = np.abs(x - exact[name + "_x"]).max()
        assert (st == 0).all() and err < FORCE_TOL, (name, err)
    for name, state in zip(("n12", "n6"), golden_states()):
        tau, grf, st = ctx.balance_solve_host(state)
        assert (st == 0).all()
        B = grf.shape[0]
        x = np.stack([grf[b].reshape(4, 3)[state["stance"][b] != 0].ravel() for b in range(B)])
        err = np.abs(x - exact[name + "_x"]).max()
        assert err < FORCE_TOL, (name, err)


@pytest.mark.gpu
def test_device_leg_kinematics_against_independent_urdf_evaluation(gpu, model):
    capi, ctx, torch = gpu
    g = model["gravity_in_base"]
    # base orientation whose inverse takes (0, 0, -9.8) to g: the shortest rotation between the two directions
    a, b = g / np.linalg.norm(g, axis=1, keepdims=True), np.array([0.0, 0.0, -1.0])
    w = 1.0 + a @ b
    v = np.cross(a, b)
    flip = w < 1e-9
    quat = np.concatenate([w[:, None], v], axis=1)
    quat[flip] = [0.0, 1.0, 0.0, 0.0]
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    foot, jac, grav = capi.leg_kinematics(ctx, model["q"], quat)
    assert np.abs(foot - model["foot"]).max() < 1e-13
    assert np.abs(jac.reshape(-1, 4, 3, 3) - model["jacobian"]).max() < 1e-13
    assert np.abs(grav - model["gravity_torque"]).max() < 1e-11
