"""GPU parity of the whole-body (floating-base) entries against oracle/oracle_wholebody.c.

The oracle works in link coordinates with Pluecker transforms and solves the 6 nS-variable QP with its equalities; the
device works in base coordinates with prefix/suffix sums over lanes and solves the 3 nS-variable eliminated QP with the
explicit-operator active-set method.  Tolerance on joint efforts: 1e-6, BASELINE.json's bar for the control step."""
import numpy as np
import pytest

from quadruped_locomotion_amd import synth

pytestmark = pytest.mark.gpu
TAU_TOL = 1e-6


@pytest.fixture(scope="module")
def gpu():
    import torch
    from quadruped_locomotion_amd import capi
    assert torch.cuda.is_available(), "these tests need the MI355X"
    capi.lib()
    ctx = capi.Context(device=0)
    yield capi, ctx, torch
    ctx.close()


@pytest.mark.parametrize("form", ["row", "leg"])
def test_dynamics_match_oracle(gpu, oracle, form):
    """Both layouts of the dynamics kernel (16 lanes per robot; one lane per leg) against the oracle, ragged batch
    sizes around the 4- and 16-robot wavefronts and the 8-robot staging halves."""
    capi, _, torch = gpu
    ctx = capi.Context(device=0)
    ctx.set_option(capi.OPT_DYNAMICS_FORM, capi.DYNAMICS_ROW if form == "row" else capi.DYNAMICS_LEG)
    for B in (1, 3, 7, 9, 17, 257):
        s = synth.make_wholebody_states(B, "trot")
        out = capi.wholebody_dynamics(ctx, s)
        for i in range(0, B, max(1, B // 40)):
            q, quat = s["q"][i], s["base_quat"][i]
            Rm = oracle.quat_to_matrix(quat)
            nu = np.concatenate([Rm.T @ s["base_linvel"][i], s["base_angvel"][i], s["qd"][i]])
            M0 = oracle.wb_mass_matrix(q)
            assert np.abs(out["M"][i] - M0).max() < 1e-11 * np.abs(M0).max()
            h0 = oracle.wb_nonlinear_effects(q, quat, nu, 9.81)
            assert np.abs(out["h"][i] - h0).max() < 1e-10 * max(1.0, np.abs(h0).max())
            assert np.abs(out["Jc"][i] - oracle.wb_contact_jacobian(q)).max() < 1e-12
    # any subset of the outputs, and device buffers
    s = synth.make_wholebody_states(130, "static")
    only_h = capi.wholebody_dynamics(ctx, s, want=("h",))
    full = capi.wholebody_dynamics(ctx, s)
    # (the instantiation that computes h alone may contract its multiply-adds differently from the one that computes all)
    assert only_h["M"] is None and np.abs(only_h["h"] - full["h"]).max() < 1e-12 * np.abs(full["h"]).max()
    d = capi.to_device(s)
    M = torch.zeros(130, 18, 18, dtype=torch.float64, device="cuda:0")
    capi.wholebody_dynamics_device(ctx, d, M, None, None)
    torch.cuda.synchronize()
    assert np.abs(M.cpu().numpy() - full["M"]).max() < 1e-13 * np.abs(full["M"]).max()
    ctx.close()


def test_dynamics_layouts_agree_at_scale(gpu):
    """20 001 robots (the library's own choice there is the one-lane-per-leg form): both layouts give the same M, h
    and Jc to rounding, M is symmetric, and the last, partly filled wavefront writes nothing past the batch."""
    capi, _, torch = gpu
    B = 20001
    d = capi.to_device(synth.make_wholebody_states(B, "trot"))
    out = {}
    for form in (capi.DYNAMICS_ROW, capi.DYNAMICS_LEG, capi.DYNAMICS_AUTO):
        ctx = capi.Context(device=0)
        ctx.set_option(capi.OPT_DYNAMICS_FORM, form)
        M = torch.full((B + 16, 18, 18), 7.0, dtype=torch.float64, device="cuda:0")
        h = torch.full((B + 16, 18), 7.0, dtype=torch.float64, device="cuda:0")
        J = torch.full((B + 16, 12, 18), 7.0, dtype=torch.float64, device="cuda:0")
        capi.wholebody_dynamics_device(ctx, d, M[:B], h[:B], J[:B])
        torch.cuda.synchronize()
        assert bool((M[B:] == 7.0).all()) and bool((h[B:] == 7.0).all()) and bool((J[B:] == 7.0).all())
        out[form] = (M[:B].cpu().numpy(), h[:B].cpu().numpy(), J[:B].cpu().numpy())
        ctx.close()
    ref = out[capi.DYNAMICS_ROW]
    for form in (capi.DYNAMICS_LEG, capi.DYNAMICS_AUTO):
        for a, b in zip(out[form], ref):
            assert np.abs(a - b).max() < 1e-12 * max(1.0, np.abs(b).max())
    assert np.array_equal(out[capi.DYNAMICS_LEG][0], out[capi.DYNAMICS_AUTO][0])
    M = out[capi.DYNAMICS_LEG][0]
    assert np.abs(M - M.transpose(0, 2, 1)).max() == 0.0


@pytest.mark.parametrize("gait", ["static", "trot"])
def test_solve_4096_matches_oracle(gpu, oracle, gait):
    capi, ctx, torch = gpu
    s = synth.make_wholebody_states(4096, gait)
    prm, oprm = capi.default_wholebody_params(), oracle.default_wb_params()
    tau, grf, st = capi.wholebody_solve(ctx, s, prm)
    t0, g0, s0 = oracle.wb_step_batch(s, oprm, nthreads=8)
    assert np.array_equal(st, s0)
    ok = st == 0
    assert ok.sum() > 3500
    err = np.abs(tau[ok] - t0[ok]).max(axis=1)
    assert err.max() < TAU_TOL, err.max()
    assert np.abs(grf[ok] - g0[ok]).max() < 1e-6
    assert np.all(tau[~ok] == 0) and np.all(grf[~ok] == 0)
    print(gait, "non-OK:", int((~ok).sum()), "max |dtau|:", err.max(), "median:", np.median(err))


def test_solve_with_binding_torque_limits_subsets_and_normals(gpu, oracle):
    """Torque limits tight enough to enter the active set, every stance subset, per-leg surface normals, desired joint
    accelerations."""
    capi, ctx, torch = gpu
    B = 2048
    s = synth.make_wholebody_states(B, "trot")
    rng = np.random.default_rng(8)
    s["stance"][:] = rng.integers(0, 2, (B, 4)).astype(np.uint8)
    s["stance"][:16] = [[(m >> l) & 1 for l in range(4)] for m in range(16)]
    nw = np.tile(np.array([0, 0, 1.0]), (B, 4, 1)) + 0.15 * rng.normal(size=(B, 4, 3))
    nw /= np.linalg.norm(nw, axis=2, keepdims=True)
    s["normals"] = nw.reshape(B, 12)
    s["qdd_des"] = rng.normal(scale=2.0, size=(B, 12))
    prm, oprm = capi.default_wholebody_params(), oracle.default_wb_params()
    prm.torque_limit = oprm.torque_limit = 45.0
    tau, grf, st = capi.wholebody_solve(ctx, s, prm)
    t0, g0, s0 = oracle.wb_step_batch(s, oprm, nthreads=8)
    assert np.array_equal(st, s0)
    ok = st == 0
    assert ok.sum() > B // 2
    assert np.abs(tau[ok] - t0[ok]).max() < TAU_TOL
    stance_joints = np.repeat(s["stance"].astype(bool), 3, axis=1)
    bound = np.isclose(np.abs(tau), 45.0, atol=1e-6) & stance_joints & ok[:, None]
    assert bound.any(axis=1).sum() > 100                              # the torque rows really bind
    assert np.abs(tau[ok][stance_joints[ok]]).max() <= 45.0 + 1e-6
    print("non-OK:", int((~ok).sum()), "robots with a torque row active:", int(bound.any(axis=1).sum()))


def test_solve_full_size_properties(gpu, oracle):
    """65 536 trot robots with tight torque limits: what holds at any size.  Every solved robot's forces lie in the
    friction pyramid with at least the minimal normal force, every stance torque within the limit, efforts of stance
    legs equal tau0 - J'f of the returned forces (through the dynamics entry's contact Jacobian and bias forces), swing
    legs carry the inverse-dynamics torque alone; a 512-robot sample agrees with the oracle."""
    capi, ctx, torch = gpu
    B = 65536
    s = synth.make_wholebody_states(B, "trot")
    prm, oprm = capi.default_wholebody_params(), oracle.default_wb_params()
    prm.torque_limit = oprm.torque_limit = 60.0
    d = capi.to_device(s)
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    grf = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    st = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
    capi.wholebody_solve_device(ctx, d, tau, grf, st, prm)
    Jc = torch.zeros(B, 12, 18, dtype=torch.float64, device="cuda:0")
    capi.wholebody_dynamics_device(ctx, d, None, None, Jc)
    torch.cuda.synchronize()
    tau, grf, st, Jc = tau.cpu().numpy(), grf.cpu().numpy(), st.cpu().numpy(), Jc.cpu().numpy()
    ok = st == 0
    assert ok.mean() > 0.99
    stance = s["stance"].astype(bool)
    f = grf.reshape(B, 4, 3)
    mu, fmin = oprm.friction, oprm.min_normal_force
    on = stance & ok[:, None]
    assert (f[on][:, 2] >= fmin - 1e-6).all()                              # flat ground: the normal is the base z axis
    # the pyramid's tangents follow the base yaw, so in base axes the forces lie in the cone that contains every such pyramid
    assert (np.hypot(f[on][:, 0], f[on][:, 1]) <= np.sqrt(2.0) * mu * f[on][:, 2] + 1e-6).all()
    assert (f[~stance] == 0).all()
    on_j = np.repeat(on, 3, axis=1)
    assert np.abs(tau[on_j]).max() <= 60.0 + 1e-6 and np.isclose(np.abs(tau[on_j]), 60.0, atol=1e-6).sum() > 50
    # tau = tau0 - J_leg' f: with f = 0 the solve returns tau0, so the difference of two solves isolates -J'f
    sw = {k: v.copy() for k, v in s.items()}
    sw["stance"][:] = 0
    tau0 = capi.wholebody_solve(ctx, {k: v[:4096] for k, v in sw.items()}, prm)[0]
    for b in np.nonzero(ok[:4096])[0][::16]:
        Jleg = Jc[b][:, 6:]                                               # 12 x 12, block diagonal by leg
        assert np.abs(tau[b] - (tau0[b] - Jleg.T @ grf[b])).max() < 1e-8
    idx = np.arange(0, B, B // 512)
    t0, g0, s0 = oracle.wb_step_batch({k: v[idx] for k, v in s.items()}, oprm, nthreads=8)
    assert np.array_equal(st[idx], s0)
    good = s0 == 0
    assert np.abs(tau[idx][good] - t0[good]).max() < TAU_TOL


def test_device_buffers_ragged_and_non_finite(gpu, oracle):
    capi, ctx, torch = gpu
    s = synth.make_wholebody_states(67, "trot")
    clean = capi.wholebody_solve(ctx, s)
    d = capi.to_device(s)
    tau = torch.full((67, 12), np.nan, dtype=torch.float64, device="cuda:0")
    grf = torch.full((67, 12), np.nan, dtype=torch.float64, device="cuda:0")
    st = torch.full((67,), -1, dtype=torch.int32, device="cuda:0")
    capi.wholebody_solve_device(ctx, d, tau, grf, st)
    torch.cuda.synchronize()
    assert np.array_equal(tau.cpu().numpy(), clean[0]) and np.array_equal(st.cpu().numpy(), clean[2])
    bad = {k: v.copy() for k, v in s.items()}
    bad["q"][1, 4] = np.nan
    bad["a_des"][5, 0] = np.inf
    bad["base_quat"][9] = 0.0
    got = capi.wholebody_solve(ctx, bad)
    keep = np.setdiff1d(np.arange(67), [1, 5, 9])
    assert np.array_equal(got[0][keep], clean[0][keep]) and np.array_equal(got[2][keep], clean[2][keep])


def test_keep_on_failure_applies_to_the_whole_body_step(oracle):
    """QLAMD_OPT_ON_FAILURE is a context option and WholeBodyController shares the plugin's context: a robot whose
    whole-body solve fails keeps what the caller's arrays held, host and device buffers alike."""
    import torch
    from quadruped_locomotion_amd import capi
    B = 300
    s = synth.make_wholebody_states(B, "trot")
    bad = {k: v.copy() for k, v in s.items()}
    broken = np.arange(2, B, 5)
    bad["q"][broken] = np.nan
    ctx = capi.Context()
    ref_tau, ref_grf, ref_st = capi.wholebody_solve(ctx, bad)
    failed = ref_st != 0
    assert failed[broken].all() and not failed.all()
    assert (ref_tau[failed] == 0).all()                                # default policy: zeros
    ctx.set_option(capi.OPT_ON_FAILURE, capi.ON_FAILURE_KEEP)
    rng = np.random.default_rng(5)
    pre_tau, pre_grf = rng.normal(size=(B, 12)), rng.normal(size=(B, 12))
    tau, grf = pre_tau.copy(), pre_grf.copy()
    _, _, st = capi.wholebody_solve(ctx, bad, tau=tau, grf=grf)
    assert np.array_equal(st, ref_st)
    assert np.array_equal(tau[failed], pre_tau[failed]) and np.array_equal(grf[failed], pre_grf[failed])
    assert np.array_equal(tau[~failed], ref_tau[~failed]) and np.array_equal(grf[~failed], ref_grf[~failed])
    d = capi.to_device(bad)
    dt, dg = torch.from_numpy(pre_tau).to("cuda:0"), torch.from_numpy(pre_grf).to("cuda:0")
    dst = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
    capi.wholebody_solve_device(ctx, d, dt, dg, dst)
    torch.cuda.synchronize()
    assert np.array_equal(dt.cpu().numpy()[failed], pre_tau[failed]) and np.array_equal(dt.cpu().numpy()[~failed], ref_tau[~failed])
    ctx.close()
