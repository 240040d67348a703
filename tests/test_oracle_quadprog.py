"""Oracle pinning: the plain-C Goldfarb-Idnani restatement against the golden
vectors produced by the reference's own compiled QuadProg++ (tests/tools/gen_goldens.py)
and, where oracle/_ref was built, against that solver live."""
import numpy as np
import pytest


@pytest.mark.parametrize("key,n,m", [("n12", 12, 20), ("n6", 6, 10)])
def test_force_qp_goldens(oracle, goldens, key, n, m):
    G, g0, CI, ci0 = (goldens[key + s] for s in ("_G", "_g0", "_CI", "_ci0"))
    for k in range(G.shape[0]):
        r = oracle.solve_quadprog(G[k], g0[k], None, None, CI[k], ci0[k])
        assert r["status"] == goldens[key + "_status"][k] == 0
        # same algorithm, same operation order: agreement to the last bits
        assert np.abs(r["x"] - goldens[key + "_x"][k]).max() <= 1e-12 * max(1.0, np.abs(r["x"]).max())
        assert abs(r["f"] - goldens[key + "_f"][k]) <= 1e-12 * max(1.0, abs(r["f"]))
        # the answer is feasible and the active rows are tight
        s = CI[k].T @ r["x"] + ci0[k]
        assert s.min() > -1e-6
        assert np.abs(s[r["active"]]).max() < 1e-6 if len(r["active"]) else True


def test_demo_literals(oracle, goldens):
    """qp_solver/src/main.cc:46-101: with its all-zero equality column the shipped demo
    returns (1.667,-0.333), f=0.722; without it the true optimum."""
    g = goldens
    r = oracle.solve_quadprog(g["demo_G"], g["demo_g0"], np.zeros((2, 1)), np.zeros(1), g["demo_CI"], g["demo_ci0"])
    assert np.allclose(r["x"], g["demo_x_dummy_eq"], atol=1e-14) and np.allclose(r["x"], [5 / 3, -1 / 3], atol=1e-12)
    assert abs(r["f"] - 0.7222222222222222) < 1e-12
    r = oracle.solve_quadprog(g["demo_G"], g["demo_g0"], None, None, g["demo_CI"], g["demo_ci0"])
    assert np.allclose(r["x"], g["demo_x"], atol=1e-14) and np.allclose(r["x"], [2 / 3, 4 / 3], atol=1e-12)
    assert abs(r["f"] + 8.222222222222222) < 1e-12


def test_three_variable_literal(oracle, goldens):
    """qp_solver/src/qp_solve_test.cpp:39-57 through the wrapper's sign flip
    CI = -A' (quadraticproblemsolver.cpp:164)."""
    g = goldens
    r = oracle.solve_quadprog(g["t3_H"], g["t3_g"], np.zeros((3, 1)), np.zeros(1), -g["t3_A"].T, g["t3_b"])
    assert np.allclose(r["x"], g["t3_x_dummy_eq"], atol=1e-14)
    assert (g["t3_A"] @ r["x"] <= g["t3_b"] + 1e-12).all()


def test_error_paths(oracle):
    # non-PD Hessian: the reference throws std::logic_error (QuadProg++.cc:692-699)
    r = oracle.solve_quadprog(np.array([[1.0, 2.0], [2.0, 1.0]]), np.zeros(2), None, None, np.zeros((2, 0)), np.zeros(0))
    assert r["status"] == oracle.QP_NOT_PD
    # infeasible: x >= 1 and -x >= 1  -> +inf (QuadProg++.cc:339-344)
    r = oracle.solve_quadprog(np.eye(1), np.zeros(1), None, None, np.array([[1.0, -1.0]]), np.array([-1.0, -1.0]))
    assert r["status"] == oracle.QP_INFEASIBLE and np.isinf(r["f"])
    # no constraints at all
    r = oracle.solve_quadprog(2 * np.eye(3), np.array([2.0, -4.0, 6.0]), None, None, np.zeros((3, 0)), np.zeros(0))
    assert np.allclose(r["x"], [-1, 2, -3])


def test_against_live_reference(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(7)
    for _ in range(300):
        n, m, p = rng.integers(2, 13), rng.integers(0, 21), rng.integers(0, 2)
        M = rng.normal(size=(n, n))
        G = M @ M.T + 1e-3 * np.eye(n)
        g0 = 10 * rng.normal(size=n)
        CI, ci0 = rng.normal(size=(n, m)), rng.normal(size=m) + 1.0
        CE = np.zeros((n, p)) if rng.random() < 0.5 else rng.normal(size=(n, p))
        a = oracle.solve_quadprog(G, g0, CE, np.zeros(p), CI, ci0)
        b = oracle.ref_solve_quadprog(G, g0, CE, np.zeros(p), CI, ci0)
        assert a["status"] == b["status"]
        if a["status"] == 0:
            assert np.array_equal(a["x"], b["x"]) and a["f"] == b["f"]


def test_dummy_zero_equality_is_an_equality_along_the_first_column_of_G(oracle):
    """SURVEY.md Q1 explained: the all-zero equality column the reference always passes consumes the first column of
    J = L^-T without moving x, i.e. it acts as the equality  (G e_1)'(x - x0) = 0  held at the unconstrained minimiser
    x0.  The lane-cooperative pose QP (csrc/pose_coop.hpp) relies on this: its projector starts as
    G^-1 - e_1 e_1'/G_11.  Checked here against the pinned restatement of the reference solver."""
    rng = np.random.default_rng(12)
    n_cmp, differs = 0, 0
    for _ in range(300):
        n, m = int(rng.integers(2, 9)), int(rng.integers(1, 10))
        A = rng.normal(size=(n, n)); G = A @ A.T + 0.5 * np.eye(n); g0 = rng.normal(size=n)
        CI = rng.normal(size=(n, m)); ci0 = rng.uniform(-0.5, 1.0, size=m)
        dummy = oracle.solve_quadprog(G.copy(), g0, np.zeros((n, 1)), np.zeros(1), CI, ci0)
        x0 = -np.linalg.solve(G, g0)
        g1 = G[:, 0].copy()
        equal = oracle.solve_quadprog(G.copy(), g0, g1[:, None], np.array([-g1 @ x0]), CI, ci0)
        plain = oracle.solve_quadprog(G.copy(), g0, None, None, CI, ci0)
        assert dummy["status"] == equal["status"]
        if dummy["status"] == 0:
            n_cmp += 1
            assert np.abs(dummy["x"] - equal["x"]).max() < 1e-9
            differs += plain["status"] == 0 and np.abs(dummy["x"] - plain["x"]).max() > 1e-3
    assert n_cmp > 100 and differs > 50          # and it is not the plain inequality-only optimum
