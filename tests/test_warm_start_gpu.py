"""Warm start of the balance step's active-set loop (qlamd_placement::prev_working_set / working_set): the loop starts from
the working set the caller hands in instead of the empty one.  The minimiser is unique, so whatever the set -- the robot's own
final set of the previous control step, a stale one, garbage -- efforts, forces and statuses are those of the cold start to
the solver's accuracy, and the oracle's within the north star's 1e-6."""
import ctypes as C

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


def solve(gpu, d, B, prev_ws=None, want_ws=True, order=None):
    capi, ctx, torch = gpu
    tau = torch.full((B, 12), np.nan, dtype=torch.float64, device="cuda:0")
    grf = torch.full((B, 12), np.nan, dtype=torch.float64, device="cuda:0")
    status = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
    iters = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
    ws = torch.full((B,), -1, dtype=torch.int32, device="cuda:0") if want_ws else None
    p = None if prev_ws is None else torch.from_numpy(np.ascontiguousarray(prev_ws).view(np.int32)).to("cuda:0")
    o = None if order is None else torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)).to("cuda:0")
    ctx.balance_solve_placed_device(d, tau, grf, status, order=o, iterations=iters, prev_working_set=p, working_set=ws,
                                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return (tau.cpu().numpy(), grf.cpu().numpy(), status.cpu().numpy(), iters.cpu().numpy(),
            None if ws is None else ws.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("gait,errors,B", [("static", "survey", 4096), ("trot", None, 4099), ("static", "calm", 1024), ("trot", None, 16385)])
def test_warm_start_reaches_the_cold_start_s_answer(gpu, oracle, gait, errors, B):
    capi, ctx, torch = gpu
    s = synth.make_states(B, gait, errors=errors)
    d = capi.to_device(s)
    t0, g0, s0, it0, ws0 = solve(gpu, d, B)                       # cold through the warm kernel (no set handed in)
    tp = torch.full((B, 12), np.nan, dtype=torch.float64, device="cuda:0")
    gp = torch.full((B, 12), np.nan, dtype=torch.float64, device="cuda:0")
    sp = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
    ctx.balance_solve_device(d, tp, gp, sp, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(t0, tp.cpu().numpy()) and np.array_equal(g0, gp.cpu().numpy()) and np.array_equal(s0, sp.cpu().numpy())
    assert (s0 == 0).all()
    # the working set written is the set of active rows: a bit per row, only rows of stance legs, as many as fit the variables
    stance_bits = np.zeros(B, dtype=np.uint32)
    for leg in range(4):
        stance_bits |= np.where(s["stance"][:, leg] != 0, np.uint32(0x1F << (5 * leg)), np.uint32(0))
    rows0 = ws0 & np.uint32(0xFFFFF)                 # bits 20..23: the support legs the set was reached with
    legs0 = (s["stance"].astype(bool) * (1 << np.arange(4))).sum(axis=1).astype(np.uint32)
    assert ((rows0 & ~stance_bits) == 0).all() and np.array_equal(ws0 >> 20, legs0)
    nact = np.array([bin(int(w)).count("1") for w in rows0])
    assert (nact <= 3 * s["stance"].astype(bool).sum(axis=1)).all() and nact.max() >= 4
    # (1) its own final set: nothing left to do but one selection that finds no violated row
    t1, g1, s1, it1, ws1 = solve(gpu, d, B, prev_ws=ws0)
    assert np.array_equal(s1, s0) and np.array_equal(ws1, ws0)
    assert np.abs(t1 - t0).max() < 1e-7 and np.abs(g1 - g0).max() < 1e-7
    nset = nact
    assert ((it1 - nset) <= 1).mean() > 0.99 and (it1 - nset).max() <= 3 and (it1 >= nset).all()   # installs + one selection
    # (2) the set of the state one control period earlier -- what a 400 Hz caller has
    prev = synth.next_tick_states(s, -0.0025)
    _, _, sprev, _, ws_prev = solve(gpu, capi.to_device(prev), B)
    t2, g2, s2, it2, ws2 = solve(gpu, d, B, prev_ws=ws_prev)
    assert np.array_equal(s2, s0)
    assert np.abs(t2 - t0).max() < 1e-7 and np.abs(g2 - g0).max() < 1e-7
    assert (ws2 == ws0).mean() > 0.999          # the same vertex (a degenerate one may be named by another basis)
    assert (it2 - nset).mean() < 1.5            # the passes still needed once the previous set is installed
    # (3) garbage: random sets, every row at once, rows of swing legs, both signs of a friction pair
    rng = np.random.default_rng(7)
    for junk in (rng.integers(0, 1 << 20, size=B, dtype=np.uint32), np.full(B, (1 << 20) - 1, dtype=np.uint32),
                 np.full(B, 0b00110_00110_00110_00110, dtype=np.uint32), rng.integers(0, 1 << 32, size=B, dtype=np.uint64).astype(np.uint32)):
        # a set unrelated to the robot's state may fail the final check: with QLAMD_OPT_WARM_FALLBACK 0 that is reported (outputs as
        # for a failed solve, working set 0: the next step starts cold) -- never a wrong answer; by default (1) the robot is solved
        # again cold inside the same launch and nothing is ever reported (tests/test_trajectory_gpu.py has the details)
        ctx.set_option(capi.OPT_WARM_FALLBACK, 0)
        try:
            t3, g3, s3, it3, ws3 = solve(gpu, d, B, prev_ws=junk)
        finally:
            ctx.set_option(capi.OPT_WARM_FALLBACK, 1)
        rej = s3 == capi.STATUS_WARM_REJECTED
        assert np.array_equal(s3[~rej], s0[~rej]) and rej.mean() < 0.01
        assert (t3[rej] == 0.0).all() and (ws3[rej] == 0).all()
        assert np.abs(t3[~rej] - t0[~rej]).max() < 1e-6 and np.abs(g3[~rej] - g0[~rej]).max() < 1e-6
        t3, g3, s3, it3, ws3 = solve(gpu, d, B, prev_ws=junk)
        assert np.array_equal(s3, s0) and np.abs(t3 - t0).max() < 1e-6 and np.abs(g3 - g0).max() < 1e-6
    # (4) against the oracle, and together with a placement
    to, go, so = oracle.balance_batch(s, nthreads=8)
    order = rng.permutation(B).astype(np.int32)
    t4, g4, s4, _, _ = solve(gpu, d, B, prev_ws=ws_prev, order=order)
    # a placement never changes the answer.  (Cold, not a bit of it: tests/test_placement_gpu.py.  Warm-started, the form in which
    # the previous set is installed -- by rounds of four rows or row by row -- is chosen per wavefront, by the largest set among
    # its four robots: the same minimiser, rounded along another path.)
    assert np.abs(t4 - t2).max() < 1e-7 and np.array_equal(s4, s2)
    assert np.array_equal(s2, so) and np.abs(t2 - to).max() < TAU_TOL and np.abs(g2 - go).max() < 1e-6


def test_the_callers_loop_at_the_baseline_batch(gpu, oracle):
    """BASELINE configs[3]'s global batch on one GPU, as a caller would run it: 65 536 trot robots, five control steps of the
    placed + warm-started loop (the 168-register form of the kernel, both forms of the QP, the sorted placement by class made
    by sixteen shadow wavefronts, working sets updated in place), odd steps on the states one control period later -- every
    step's efforts and statuses against the oracle's for the states it ran on."""
    capi, ctx, torch = gpu
    B = 65536
    a = synth.make_states(B, "trot")
    b = synth.next_tick_states(a, 0.0025)
    states, devs = [a, b], [capi.to_device(a), capi.to_device(b)]
    want = [oracle.balance_batch(s, nthreads=16) for s in states]
    order = [torch.arange(B, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    iters = [torch.zeros(B, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    ws = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for k in range(5):
        tau = torch.full((B, 12), np.nan, dtype=torch.float64, device="cuda:0")
        status = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
        ctx.balance_solve_placed_device(devs[k & 1], tau, None, status, order=order[k & 1], iterations=iters[k & 1],
                                        prev_iterations=iters[(k - 1) & 1], next_order=order[(k + 1) & 1],
                                        policy=capi.PLACEMENT_AUTO, prev_working_set=ws, working_set=ws, stream=stream)
        torch.cuda.synchronize()
        t0, _, s0 = want[k & 1]
        st = status.cpu().numpy()
        assert np.array_equal(st, s0), (k, int((st != s0).sum()))
        ok = s0 == 0
        assert np.abs(tau.cpu().numpy()[ok] - t0[ok]).max() < TAU_TOL, k
        nxt = order[(k + 1) & 1].cpu().numpy()
        assert np.array_equal(np.sort(nxt), np.arange(B)), k    # a permutation, whatever the hints were
        if k >= 2:  # made from real counts: robots on more than two legs first, each class hardest first
            legs = states[k & 1]["stance"].astype(bool).sum(1)
            cls = (legs[nxt] <= 2).astype(int)
            assert (np.diff(cls) >= 0).all(), k
            cnt = np.clip(iters[(k - 1) & 1].cpu().numpy(), 0, 23)[nxt]
            for c in (0, 1):
                assert (np.diff(cnt[cls == c]) <= 0).all(), (k, c)


def test_warm_start_arguments(gpu):
    capi, ctx, torch = gpu
    B = 64
    s = synth.make_states(B, "trot")
    d = capi.to_device(s)
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    ws = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    # one array for both: updated in place (a robot's set is read and written by its own lanes only)
    ws2 = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    ctx.balance_solve_placed_device(d, tau, None, status, working_set=ws2)
    ctx.balance_solve_placed_device(d, tau, None, status, prev_working_set=ws, working_set=ws)
    ctx.balance_solve_placed_device(d, tau, None, status, prev_working_set=ws, working_set=ws)
    torch.cuda.synchronize()
    assert torch.equal(ws, ws2) and (status == 0).all()
    # host memory: refused (a host-buffer call is bound by its copies)
    sb, keep = capi.StateBatch(), []
    for key, field, k in capi.FIELD_OF_KEY:
        a = np.ascontiguousarray(s[key], dtype=np.float64).reshape(B, k)
        keep.append(a)
        setattr(sb, field, a.ctypes.data)
    st8 = np.ascontiguousarray(s["stance"], dtype=np.uint8)
    sb.support_leg = st8.ctypes.data
    h_tau, h_st, h_ws = np.zeros((B, 12)), np.zeros(B, np.int32), np.zeros(B, np.uint32)
    pl = capi.Placement(None, None, None, None, 0, None, h_ws.ctypes.data)
    rc = capi.lib().qlamd_balance_solve_placed_batch(ctx._h, C.byref(sb), B, C.byref(pl), h_tau.ctypes.data, None, h_st.ctypes.data,
                                                     capi.MEM_HOST, None)
    assert rc == capi.ERR_INVALID_ARGUMENT
    # the one-lane kernels know neither
    ctx.set_robots_per_wave(64)
    try:
        with pytest.raises(capi.QlamdError):
            ctx.balance_solve_placed_device(d, tau, None, status, working_set=ws)
    finally:
        ctx.set_robots_per_wave(0)


@pytest.mark.parametrize("gait", ["trot", "static"])
def test_wholebody_step_warm_start(gpu, oracle, gait):
    """The whole-body step through qlamd_place_next_call with 64-bit working sets ([B][2] words: friction, minimal force and the
    two torque bounds of every joint): from its own final set, from the set of the state one period earlier (joint angles and
    velocities moved by their own rates), from garbage -- statuses and efforts as the cold start's, and the oracle's."""
    capi, ctx, torch = gpu
    B = 4096
    s = synth.make_wholebody_states(B, gait)
    d = capi.to_device(s)
    stream = torch.cuda.current_stream().cuda_stream

    def run(dd, prev=None, want=True):
        tau = torch.full((B, 12), np.nan, dtype=torch.float64, device="cuda:0")
        grf = torch.full((B, 12), np.nan, dtype=torch.float64, device="cuda:0")
        st = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
        it = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
        ws = torch.zeros(B, 2, dtype=torch.int32, device="cuda:0") if want else None
        p = None if prev is None else torch.from_numpy(np.ascontiguousarray(prev).view(np.int32).reshape(B, 2)).to("cuda:0")
        pl = capi.Placement(None, it.data_ptr(), None, None, 0, None if p is None else p.data_ptr(), None if ws is None else ws.data_ptr())
        assert capi.lib().qlamd_place_next_call(ctx._h, C.byref(pl)) == 0
        capi.wholebody_solve_device(ctx, dd, tau, grf, st, stream=stream)
        torch.cuda.synchronize()
        return (tau.cpu().numpy(), grf.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(),
                None if ws is None else ws.cpu().numpy().view(np.uint32).reshape(B, 2).copy().view(np.uint64).reshape(B))
    tau0 = torch.full((B, 12), np.nan, dtype=torch.float64, device="cuda:0")
    grf0 = torch.full((B, 12), np.nan, dtype=torch.float64, device="cuda:0")
    st0 = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
    capi.wholebody_solve_device(ctx, d, tau0, grf0, st0, stream=stream)
    torch.cuda.synchronize()
    t0, g0, s0 = tau0.cpu().numpy(), grf0.cpu().numpy(), st0.cpu().numpy()
    tc, gc, sc, itc, wsc = run(d)
    assert np.array_equal(tc, t0) and np.array_equal(sc, s0)      # no set handed in: the cold start, bit for bit
    rows_c = wsc & np.uint64((1 << 44) - 1)           # bits 44..47: the support legs the set was reached with
    assert (wsc < (1 << 48)).all() and (rows_c != 0).sum() > (B // 8 if gait == "trot" else 0)   # (a static stance activates next to nothing)
    ok = s0 == 0
    t1, g1, s1, it1, ws1 = run(d, prev=wsc)
    assert np.array_equal(s1, s0) and np.array_equal(ws1[ok], wsc[ok])
    assert np.abs(t1[ok] - t0[ok]).max() < 1e-7 and np.abs(g1[ok] - g0[ok]).max() < 1e-7
    nset = np.array([bin(int(w)).count("1") for w in rows_c])
    assert ((it1 - nset)[ok] <= 1).mean() > 0.98
    earlier = dict(s)
    earlier["q"] = s["q"] - 0.0025 * s["qd"]
    _, _, _, _, wsp = run(capi.to_device(earlier))
    t2, g2, s2, it2, ws2 = run(d, prev=wsp)
    assert np.array_equal(s2, s0) and np.abs(t2[ok] - t0[ok]).max() < 1e-7      # (equal statuses: nothing rejected)
    rng = np.random.default_rng(3)
    sparse = np.zeros(B, dtype=np.uint64)           # random sets that could be working sets: up to three rows per leg
    for leg in range(4):
        for _ in range(3):
            sparse |= np.where(rng.random(B) < 0.6, np.uint64(1) << (np.uint64(11 * leg) + rng.integers(0, 11, size=B).astype(np.uint64)), np.uint64(0))
    for junk in (rng.integers(0, 1 << 44, size=B, dtype=np.uint64), np.full(B, (1 << 44) - 1, dtype=np.uint64), sparse):
        ctx.set_option(capi.OPT_WARM_FALLBACK, 0)     # rejections reported ...
        try:
            t3, g3, s3, _, ws3 = run(d, prev=junk)
        finally:
            ctx.set_option(capi.OPT_WARM_FALLBACK, 1)
        rej = s3 == capi.STATUS_WARM_REJECTED
        good = ok & ~rej
        assert np.array_equal(s3[~rej], s0[~rej]) and rej.mean() < 0.02 and (ws3[rej] == 0).all()
        assert np.abs(t3[good] - t0[good]).max() < 1e-6, np.abs(t3[good] - t0[good]).max()
        t3, g3, s3, _, ws3 = run(d, prev=junk)        # ... or, by default, solved again cold by the same launch
        assert np.array_equal(s3, s0) and np.abs(t3[ok] - t0[ok]).max() < 1e-6
        assert np.array_equal(t3[rej], tc[rej]) and np.array_equal(ws3[rej & ok], wsc[rej & ok])
    to, go, so = oracle.wb_step_batch(s, nthreads=8)
    assert np.array_equal(s2, so) and np.abs(t2[so == 0] - to[so == 0]).max() < TAU_TOL
    # the second attempt at will (QLAMD_OPT_WARM_FALLBACK 2): every robot that ends with a non-empty set is solved again cold by
    # the same launch -- the cold step's answer (to rounding: the second attempt is the same source compiled as a function of
    # its own), its working set back to 0, and the context counts it
    ctx.set_option(capi.OPT_WARM_FALLBACK, 2)
    try:
        before = ctx.counter(capi.COUNTER_WARM_RETRIES)
        t4, g4, s4, _, ws4 = run(d, prev=wsc)
        n4 = ctx.counter(capi.COUNTER_WARM_RETRIES) - before
    finally:
        ctx.set_option(capi.OPT_WARM_FALLBACK, 1)
    again = ok & (rows_c != 0) & (ws4 == 0)
    assert n4 == again.sum() and again.sum() > 0.9 * (ok & (rows_c != 0)).sum()
    assert np.array_equal(s4, s0) and np.abs(t4[again] - t0[again]).max() < 1e-9 and np.abs(g4[again] - g0[again]).max() < 1e-9
    assert np.abs(t4[ok] - t0[ok]).max() < 1e-7
    # the dense entries start cold: a working set handed to them is refused
    ws = torch.zeros(B, 2, dtype=torch.int32, device="cuda:0")
    pl = capi.Placement(None, None, None, None, 0, None, ws.data_ptr())
    assert capi.lib().qlamd_place_next_call(ctx._h, C.byref(pl)) == 0
    x, st = torch.zeros(8, 12, dtype=torch.float64, device="cuda:0"), torch.zeros(8, dtype=torch.int32, device="cuda:0")
    G = torch.eye(12, dtype=torch.float64, device="cuda:0").repeat(8, 1, 1).contiguous()
    g = torch.zeros(8, 12, dtype=torch.float64, device="cuda:0")
    rc = capi.lib().qlamd_qp_solve_batch(ctx._h, 12, 0, 0, G.data_ptr(), g.data_ptr(), None, None, None, None, 8, x.data_ptr(), None,
                                         st.data_ptr(), capi.MEM_DEVICE, None)
    assert rc == capi.ERR_INVALID_ARGUMENT
