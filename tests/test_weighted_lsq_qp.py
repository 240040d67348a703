"""qlamd_weighted_lsq_qp_batch: the argument list of ooqpei::QuadraticProblemFormulation::solve(A, S, b, W, C, c, D, d, f, x)
(call sites ContactForceDistribution.cpp:367,490) -- min (Ax-b)'S(Ax-b) + x'Wx  s.t. Cx = c, d <= Dx <= f.

OOQP / ooqpei are absent (parity unpinned at that boundary): the target is the unique minimiser.  It is pinned from two
sides: the committed answers of the reference's own compiled QuadProg++ on the 256 golden force problems (tests/golden/
qp_goldens.npz) with their 40-digit minimisers (force_qp_exact.npz), and oracle_weighted_lsq_qp, which eliminates the
equalities through a null-space basis (a different route from the device's row-by-row projection)."""
import numpy as np
import pytest

from quadruped_locomotion_amd import synth

NO_BOUND = 1.7976931348623157e308


def golden_items():
    """(state dict, index) of the 256 golden force problems, in the order of tests/tools/gen_goldens.py."""
    trot = synth.make_states(4096, "trot", seed=synth.SEED)
    static = synth.make_states(512, "static", seed=synth.SEED)
    four = [(static, i) for i in range(64)] + [(trot, i) for i in range(4096) if trot["stance"][i].sum() == 4][:64]
    two = [(trot, i) for i in range(4096) if trot["stance"][i].sum() == 2][:128]
    return four, two


def lsq_form(O, s, i):
    stance = s["stance"][i]
    legs = [l for l in range(4) if stance[l]]
    Rm = O.quat_to_matrix(s["base_quat"][i])
    rf = np.array([O.leg_fk(l, s["q"][i][3 * l:3 * l + 3])[0] for l in legs])
    yB = Rm.T @ np.array([0.0, 1.0, 0.0])
    nB = Rm.T @ (Rm @ np.array([0.0, 0.0, 1.0]))
    t1 = np.cross(nB, yB); t1 /= np.linalg.norm(t1)
    t2 = np.cross(nB, t1); t2 /= np.linalg.norm(t2)
    nS = len(legs)
    return O.force_lsq_assemble(rf, O.virtual_wrench(s, i), np.tile(nB, (nS, 1)), np.tile(t1, (nS, 1)), np.tile(t2, (nS, 1)))


def stacked(O, items):
    probs = [lsq_form(O, s, i) for s, i in items]
    return [np.stack([p[k] for p in probs]) for k in range(7)]


def random_problems(rng, B, n, k, p, m, dependent_rows=False):
    A = rng.normal(size=(B, k, n))
    S = rng.uniform(0.5, 10.0, size=(B, k))
    b = 5 * rng.normal(size=(B, k))
    W = rng.uniform(1e-3, 1e-1, size=(B, n))
    x_feas = rng.normal(size=(B, n))                       # a point every problem admits
    C = rng.normal(size=(B, p, n))
    if dependent_rows and p >= 3:
        C[:, 1] = 0.0                                      # an all-zero row
        C[:, 2] = 1.5 * C[:, 0]                            # a multiple of the first
    c = np.einsum("bpn,bn->bp", C, x_feas)
    D = rng.normal(size=(B, m, n))
    mid = np.einsum("bmn,bn->bm", D, x_feas)
    d = mid - rng.uniform(0.0, 1.0, size=(B, m))
    f = mid + rng.uniform(0.0, 1.0, size=(B, m))
    d[rng.random((B, m)) < 0.3] = -NO_BOUND
    f[rng.random((B, m)) < 0.3] = NO_BOUND
    f[:, ::5] = np.inf                                     # infinity works like DBL_MAX
    return A, S, b, W, C, c, D, d, f


def kkt_check(A, S, b, W, C, c, D, d, f, x, tol=1e-6):
    """First-order optimality of x for one problem, by a least-squares fit of the multipliers of the rows that hold."""
    g = 2 * (A.T @ (S * (A @ x - b)) + W * x)
    assert np.abs(C @ x - c).max(initial=0.0) < tol
    Dx = D @ x
    lo, up = d > -NO_BOUND, f < NO_BOUND
    assert (Dx[lo] >= d[lo] - tol).all() and (Dx[up] <= f[up] + tol).all()
    rows, signs = [r for r in C], [0] * len(C)
    for r in range(len(D)):
        if lo[r] and Dx[r] - d[r] < 1e-5:
            rows.append(D[r]); signs.append(+1)
        elif up[r] and f[r] - Dx[r] < 1e-5:
            rows.append(-D[r]); signs.append(+1)
    if not rows:
        assert np.abs(g).max() < tol * max(1.0, np.abs(b).max())
        return
    N = np.array(rows).T
    lam, *_ = np.linalg.lstsq(N, g, rcond=None)
    assert np.abs(N @ lam - g).max() < 1e-5 * max(1.0, np.abs(g).max())
    assert all(l > -1e-6 * max(1.0, np.abs(lam).max()) for l, sg in zip(lam, signs) if sg)


# ---------------------------------------------------------------------------------------------- oracle (CPU)
def test_oracle_lsq_form_is_the_golden_problem_and_reaches_the_reference_answers(oracle, goldens):
    four, two = golden_items()
    for key, items in (("n12", four), ("n6", two)):
        for kk, (s, i) in enumerate(items[::4]):
            k = 4 * kk
            A, S, b, W, D, d, f = lsq_form(oracle, s, i)
            # the same problem the golden file holds in QuadProg++ form (G = A'SA + W, g0 = -A'Sb, CI = D', ci0 = -d)
            assert np.allclose(A.T @ np.diag(S) @ A + np.diag(W), goldens[key + "_G"][k], rtol=0, atol=1e-12)
            assert np.allclose(-A.T @ (S * b), goldens[key + "_g0"][k], rtol=0, atol=1e-9)
            assert np.array_equal(D.T, goldens[key + "_CI"][k]) and np.array_equal(-d, goldens[key + "_ci0"][k])
            assert (f == NO_BOUND).all()
            x, st = oracle.weighted_lsq_qp(A, S, b, W, None, None, D, d, f)
            assert st == 0 and np.abs(x - goldens[key + "_x"][k]).max() < 2e-8
            # the reference's two passes (ContactForceDistribution.cpp:364-381,490): zero rows, then pinned to the first
            n = A.shape[1]
            x1, st1 = oracle.weighted_lsq_qp(A, S, b, W, np.zeros((n, n)), np.zeros(n), D, d, f)
            x2, st2 = oracle.weighted_lsq_qp(A, S, b, W, np.eye(n), x1, D, d, f)
            assert st1 == 0 and st2 == 0 and np.abs(x1 - x).max() < 1e-12 and np.abs(x2 - x1).max() < 1e-12


def test_oracle_general_problems_satisfy_the_optimality_conditions(oracle):
    rng = np.random.default_rng(3)
    for n, k, p, m, dep in ((12, 6, 0, 20, False), (12, 12, 3, 24, True), (6, 6, 2, 10, False), (9, 4, 5, 7, True), (3, 3, 3, 4, False)):
        P = random_problems(rng, 24, n, k, p, m, dep)
        for i in range(24):
            x, st = oracle.weighted_lsq_qp(*[a[i] for a in P])
            assert st == 0
            kkt_check(*[a[i] for a in P], x)
    # inconsistent equalities, an empty feasible set, no definiteness
    A, S, b, W = np.eye(2), np.ones(2), np.zeros(2), np.full(2, 0.1)
    assert oracle.weighted_lsq_qp(A, S, b, W, np.array([[1.0, 0], [1.0, 0]]), np.array([1.0, 2.0]))[1] == 1
    assert oracle.weighted_lsq_qp(A, S, b, W, None, None, np.array([[1.0, 0], [1.0, 0]]), np.array([2.0, -NO_BOUND]), np.array([NO_BOUND, 1.0]))[1] == 1
    assert oracle.weighted_lsq_qp(np.array([[1.0, 0.0]]), np.ones(1), np.zeros(1), np.array([0.1, -0.1]))[1] == 2


# ---------------------------------------------------------------------------------------------- device
@pytest.fixture(scope="module")
def gpu():
    import torch
    from quadruped_locomotion_amd import capi
    assert torch.cuda.is_available(), "these tests need the MI355X"
    capi.lib()
    ctx = capi.Context(device=0)
    yield capi, ctx, torch
    ctx.close()


@pytest.mark.gpu
def test_golden_states_through_the_reference_two_pass_sequence(gpu, oracle, goldens):
    """The 256 golden states: the reference's literal sequence -- solve with C = 0 (3 nS zero rows), then C = I, c = x1 --
    returns x2 = x1 = what the balance kernel's single solve distributes, = the reference's own QuadProg++
    answers, = the 40-digit minimisers."""
    import os
    capi, ctx, torch = gpu
    exact = np.load(os.path.join(os.path.dirname(__file__), "golden", "force_qp_exact.npz"))
    four, two = golden_items()
    for key, items in (("n12", four), ("n6", two)):
        A, S, b, W, D, d, f = stacked(oracle, items)
        B, n = A.shape[0], A.shape[2]
        x0, st0 = capi.weighted_lsq_qp(ctx, A, S, b, W, None, None, D, d, f)
        x1, st1 = capi.weighted_lsq_qp(ctx, A, S, b, W, np.zeros((B, n, n)), np.zeros((B, n)), D, d, f)
        x2, st2 = capi.weighted_lsq_qp(ctx, A, S, b, W, np.tile(np.eye(n), (B, 1, 1)), x1, D, d, f)
        assert (st0 == 0).all() and (st1 == 0).all() and (st2 == 0).all()
        assert np.array_equal(x1, x0)                                   # rows that are skipped leave no trace
        assert np.abs(x2 - x1).max() < 1e-9
        assert np.abs(x1 - goldens[key + "_x"]).max() < 1e-7
        xs = [k for k in exact.files if k.startswith(key) and k.endswith("_x")]
        assert xs, exact.files
        assert np.abs(x1 - exact[xs[0]]).max() < 1e-7
        # the balance kernel on the same states: its contact forces are this x
        states = {k: np.stack([s[k][i] for s, i in items]) for k in items[0][0]}
        tau, grf, st = ctx.balance_solve_host(states)
        assert (st == 0).all()
        legs = np.repeat(states["stance"].astype(bool), 3, axis=1)
        assert np.abs(grf[legs].reshape(B, n) - x1).max() < 1e-7


@pytest.mark.gpu
def test_general_problems_match_the_oracle(gpu, oracle):
    capi, ctx, torch = gpu
    rng = np.random.default_rng(17)
    for n, k, p, m, dep in ((12, 6, 0, 20, False), (12, 12, 3, 24, True), (6, 6, 2, 10, False), (9, 4, 5, 7, True),
                            (3, 3, 3, 4, False), (12, 6, 12, 20, False), (1, 1, 0, 1, False), (7, 12, 0, 0, False)):
        B = 67
        P = random_problems(rng, B, n, k, p, m, dep)
        args = [a if a.shape[1] else None for a in P[:4]] + [P[4] if p else None, P[5] if p else None] + \
               [P[6] if m else None, P[7] if m else None, P[8] if m else None]
        x, st = capi.weighted_lsq_qp(ctx, *args)
        for i in range(B):
            xo, so = oracle.weighted_lsq_qp(*[None if a is None else a[i] for a in args])
            assert so == st[i] == 0, (n, k, p, m, i, so, st[i])
            assert np.abs(x[i] - xo).max() < 1e-7 * max(1.0, np.abs(xo).max()), (n, k, p, m, i)
        # device buffers give the same bits
        dargs = [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in args]
        dx = torch.zeros(B, n, dtype=torch.float64, device="cuda:0")
        dst = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
        capi.weighted_lsq_qp(ctx, *dargs, memory=capi.MEM_DEVICE, out=(dx, dst))
        torch.cuda.synchronize()
        assert np.array_equal(dx.cpu().numpy(), x) and np.array_equal(dst.cpu().numpy(), st)


@pytest.mark.gpu
def test_failures_are_reported_per_problem(gpu, oracle):
    capi, ctx, torch = gpu
    A, S, b, W = np.tile(np.eye(2), (4, 1, 1)), np.ones((4, 2)), np.zeros((4, 2)), np.full((4, 2), 0.1)
    C = np.tile(np.array([[1.0, 0.0], [1.0, 0.0]]), (4, 1, 1))
    c = np.array([[1.0, 1.0], [1.0, 2.0], [0.5, 0.5], [0.0, 3.0]])       # rows 1 and 3 contradict themselves
    x, st = capi.weighted_lsq_qp(ctx, A, S, b, W, C, c)
    assert list(st) == [0, capi.STATUS_INFEASIBLE, 0, capi.STATUS_INFEASIBLE]
    assert np.allclose(x[0], [1.0, 0.0]) and np.allclose(x[2], [0.5, 0.0])
    # an empty feasible set between two bounds; an indefinite weight
    D = np.tile(np.array([[1.0, 0.0], [1.0, 0.0]]), (4, 1, 1))
    d = np.tile(np.array([2.0, -NO_BOUND]), (4, 1)); f = np.tile(np.array([NO_BOUND, 1.0]), (4, 1))
    d[1] = [-5.0, -NO_BOUND]
    x, st = capi.weighted_lsq_qp(ctx, A, S, b, W, None, None, D, d, f)
    assert list(st) == [capi.STATUS_INFEASIBLE, 0, capi.STATUS_INFEASIBLE, capi.STATUS_INFEASIBLE]
    Wn = W.copy(); Wn[2, 1] = -2.0
    x, st = capi.weighted_lsq_qp(ctx, A, S, b, Wn)
    assert list(st) == [0, 0, capi.STATUS_NOT_PD, 0]
    with pytest.raises(capi.QlamdError):
        capi.weighted_lsq_qp(ctx, np.zeros((1, 13, 2)), np.ones((1, 13)), np.zeros((1, 13)), np.ones((1, 2)))
