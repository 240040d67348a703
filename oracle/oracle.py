"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes access to oracle/liboracle.so (our plain-C restatement of the reference
algorithms) and, where it was built, oracle/_ref/libquadprog_ref.so (the
reference's own QuadProg++ compiled from /root/reference).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

QP_OK, QP_INFEASIBLE, QP_NOT_PD, QP_BAD_DELETE = 0, 1, 2, 3


class BalanceParams(C.Structure):
    _fields_ = [
        ("kp_trans", C.c_double * 3), ("kd_trans", C.c_double * 3), ("kff_trans", C.c_double * 3),
        ("kp_rot", C.c_double * 3), ("kd_rot", C.c_double * 3), ("kff_rot", C.c_double * 3),
        ("force_weights", C.c_double * 6),
        ("regularizer", C.c_double), ("friction", C.c_double), ("min_normal_force", C.c_double),
        ("torque_limit", C.c_double), ("torso_mass", C.c_double), ("leg_mass", C.c_double * 4),
        ("gravity", C.c_double), ("grav_comp_percentage", C.c_double),
        ("com_in_base", C.c_double * 3), ("hip_in_base", (C.c_double * 3) * 4),
    ]


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/qp_solver") and not os.path.exists(os.path.join(_HERE, "_ref", "libquadprog_ref.so")):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_solve_quadprog.restype = C.c_int
        _lib.oracle_leg_potential.restype = C.c_double
    return _lib


def ref_lib():
    """The reference's own compiled QuadProg++ or None (absent on the GPU box
    unless the prebuilt oracle/_ref/ travelled with the snapshot)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libquadprog_ref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
        _ref.ref_solve_quadprog.restype = C.c_int
    return _ref


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def default_params():
    p = BalanceParams()
    lib().oracle_balance_default_params(C.byref(p))
    return p


def _qp_args(G, g0, CE, ce0, CI, ci0):
    G = np.array(G, dtype=np.float64, order="C")
    n = G.shape[0]
    g0 = np.ascontiguousarray(g0, dtype=np.float64)
    CE = np.zeros((n, 0)) if CE is None else np.ascontiguousarray(CE, dtype=np.float64).reshape(n, -1)
    ce0 = np.zeros(0) if ce0 is None else np.ascontiguousarray(ce0, dtype=np.float64)
    CI = np.zeros((n, 0)) if CI is None else np.ascontiguousarray(CI, dtype=np.float64).reshape(n, -1)
    ci0 = np.zeros(0) if ci0 is None else np.ascontiguousarray(ci0, dtype=np.float64)
    return n, CE.shape[1], CI.shape[1], G, g0, CE, ce0, CI, ci0


def solve_quadprog(G, g0, CE=None, ce0=None, CI=None, ci0=None):
    """min 1/2 x'Gx + g0'x  s.t. CE'x+ce0=0, CI'x+ci0>=0 (QuadProg++.h:11-14).
    Returns dict(x, f, status, active, iters)."""
    n, p, m, G, g0, CE, ce0, CI, ci0 = _qp_args(G, g0, CE, ce0, CI, ci0)
    x = np.zeros(n)
    f = C.c_double()
    active = np.zeros(m + p + 1, dtype=np.int32)
    nact, iters = C.c_int(), C.c_int()
    st = lib().oracle_solve_quadprog(n, p, m, G.ctypes.data_as(_dp), g0.ctypes.data_as(_dp),
                                     CE.ctypes.data_as(_dp), ce0.ctypes.data_as(_dp),
                                     CI.ctypes.data_as(_dp), ci0.ctypes.data_as(_dp),
                                     x.ctypes.data_as(_dp), C.byref(f), active.ctypes.data_as(_ip),
                                     C.byref(nact), C.byref(iters))
    return dict(x=x, f=f.value, status=st, active=active[:nact.value].copy(), iters=iters.value)


def ref_solve_quadprog(G, g0, CE=None, ce0=None, CI=None, ci0=None):
    """Same problem through the reference's own compiled solve_quadprog."""
    r = ref_lib()
    if r is None:
        raise RuntimeError("oracle/_ref/libquadprog_ref.so not built (needs /root/reference)")
    n, p, m, G, g0, CE, ce0, CI, ci0 = _qp_args(G, g0, CE, ce0, CI, ci0)
    x = np.zeros(n)
    f = C.c_double()
    st = r.ref_solve_quadprog(n, p, m, G.ctypes.data_as(_dp), g0.ctypes.data_as(_dp),
                              CE.ctypes.data_as(_dp), ce0.ctypes.data_as(_dp),
                              CI.ctypes.data_as(_dp), ci0.ctypes.data_as(_dp),
                              x.ctypes.data_as(_dp), C.byref(f))
    return dict(x=x, f=f.value, status=st)


def leg_fk(leg, q):
    q, qp = _d(q)
    p = np.zeros(3)
    R = np.zeros(9)
    lib().oracle_leg_fk(int(leg), qp, p.ctypes.data_as(_dp), R.ctypes.data_as(_dp))
    return p, R.reshape(3, 3)


def leg_jacobian(leg, q):
    q, qp = _d(q)
    J = np.zeros(9)
    lib().oracle_leg_jacobian(int(leg), qp, J.ctypes.data_as(_dp))
    return J.reshape(3, 3)


def leg_gravity(leg, q, g):
    q, qp = _d(q)
    g, gp = _d(g)
    G = np.zeros(3)
    lib().oracle_leg_gravity(int(leg), qp, gp, G.ctypes.data_as(_dp))
    return G


def leg_potential(leg, q, g):
    q, qp = _d(q)
    g, gp = _d(g)
    return lib().oracle_leg_potential(int(leg), qp, gp)


def quat_to_matrix(q):
    q, qp = _d(q)
    R = np.zeros(9)
    lib().oracle_quat_to_matrix(qp, R.ctypes.data_as(_dp))
    return R.reshape(3, 3)


def quat_box_minus(a, b):
    a, ap = _d(a)
    b, bp = _d(b)
    o = np.zeros(3)
    lib().oracle_quat_box_minus(ap, bp, o.ctypes.data_as(_dp))
    return o


STATE_FIELDS = (("q", 12), ("base_pos", 3), ("base_quat", 4), ("base_linvel", 3), ("base_angvel", 3),
                ("des_pos", 3), ("des_quat", 4), ("des_linvel", 3), ("des_angvel", 3))


def balance_step(state, i=0, params=None, normals_world=None):
    """One robot `i` of a state dict (fields [B][k] + 'stance' uint8 [B][4])."""
    prm = params or default_params()
    ptrs, keep = [], []
    for name, k in STATE_FIELDS:
        a, p = _d(np.asarray(state[name]).reshape(-1, k)[i])
        keep.append(a)
        ptrs.append(p)
    st = np.ascontiguousarray(np.asarray(state["stance"]).reshape(-1, 4)[i], dtype=np.uint8)
    nw = None
    if normals_world is not None:
        nwa, nw = _d(np.asarray(normals_world).reshape(-1, 12)[i])
        keep.append(nwa)
    tau, tau_raw, grf, wrench = np.zeros(12), np.zeros(12), np.zeros(12), np.zeros(6)
    iters, nact = C.c_int(), C.c_int()
    status = lib().oracle_balance_step(C.byref(prm), *ptrs, st.ctypes.data_as(C.POINTER(C.c_uint8)), nw,
                                       tau.ctypes.data_as(_dp), tau_raw.ctypes.data_as(_dp),
                                       grf.ctypes.data_as(_dp), wrench.ctypes.data_as(_dp),
                                       C.byref(iters), C.byref(nact))
    return dict(tau=tau, tau_raw=tau_raw, grf=grf, wrench=wrench, status=status,
                iters=iters.value, n_active=nact.value)


def balance_batch(state, params=None, normals_world=None, nthreads=1):
    """Whole batch; returns tau [B,12], grf [B,12], status [B]."""
    prm = params or default_params()
    B = int(np.asarray(state["q"]).reshape(-1, 12).shape[0])
    ptrs, keep = [], []
    for name, k in STATE_FIELDS:
        a, p = _d(np.asarray(state[name]).reshape(B, k))
        keep.append(a)
        ptrs.append(p)
    st = np.ascontiguousarray(np.asarray(state["stance"]).reshape(B, 4), dtype=np.uint8)
    nw = None
    if normals_world is not None:
        nwa, nw = _d(np.asarray(normals_world).reshape(B, 12))
        keep.append(nwa)
    tau, grf = np.zeros((B, 12)), np.zeros((B, 12))
    status = np.zeros(B, dtype=np.int32)
    lib().oracle_balance_batch(C.byref(prm), C.c_int64(B), *ptrs, st.ctypes.data_as(C.POINTER(C.c_uint8)), nw,
                               tau.ctypes.data_as(_dp), grf.ctypes.data_as(_dp),
                               status.ctypes.data_as(C.POINTER(C.c_int32)), int(nthreads))
    return tau, grf, status


def balance_batch_repeat(state, passes, nthreads, params=None):
    """`passes` passes over the batch inside one parallel region (bench.py cpu_baseline); returns the wall seconds
    measured in C between two barriers."""
    prm = params or default_params()
    B = int(np.asarray(state["q"]).reshape(-1, 12).shape[0])
    ptrs, keep = [], []
    for name, k in STATE_FIELDS:
        a, p = _d(np.asarray(state[name]).reshape(B, k))
        keep.append(a)
        ptrs.append(p)
    st = np.ascontiguousarray(np.asarray(state["stance"]).reshape(B, 4), dtype=np.uint8)
    tau, status = np.zeros((B, 12)), np.zeros(B, dtype=np.int32)
    fn = lib().oracle_balance_batch_repeat
    fn.restype = C.c_double
    return float(fn(C.byref(prm), C.c_int64(B), *ptrs, st.ctypes.data_as(C.POINTER(C.c_uint8)), tau.ctypes.data_as(_dp),
                    status.ctypes.data_as(C.POINTER(C.c_int32)), int(nthreads), int(passes)))


def virtual_wrench(state, i=0, params=None):
    prm = params or default_params()
    ptrs, keep = [], []
    for name, k in STATE_FIELDS[1:]:
        a, p = _d(np.asarray(state[name]).reshape(-1, k)[i])
        keep.append(a)
        ptrs.append(p)
    w = np.zeros(6)
    lib().oracle_virtual_wrench(C.byref(prm), *ptrs, w.ctypes.data_as(_dp))
    return w


def force_qp_assemble(r_feet, wrench, n_B, t1, t2, params=None):
    prm = params or default_params()
    r_feet = np.ascontiguousarray(r_feet, dtype=np.float64).reshape(-1, 3)
    nS = r_feet.shape[0]
    n, m = 3 * nS, 5 * nS
    G, g0, CI, ci0 = np.zeros((n, n)), np.zeros(n), np.zeros((n, m)), np.zeros(m)
    a = [_d(v) for v in (r_feet, wrench, np.reshape(n_B, (nS, 3)), np.reshape(t1, (nS, 3)), np.reshape(t2, (nS, 3)))]
    lib().oracle_force_qp_assemble(C.byref(prm), nS, a[0][1], a[1][1], a[2][1], a[3][1], a[4][1],
                                   G.ctypes.data_as(_dp), g0.ctypes.data_as(_dp),
                                   CI.ctypes.data_as(_dp), ci0.ctypes.data_as(_dp))
    return G, g0, CI, ci0


def force_lsq_assemble(r_feet, wrench, n_B, t1, t2, params=None):
    """The force problem in the reference's own (A, S, b, W, D, d, f) form (ContactForceDistribution.cpp:168-336)."""
    prm = params or default_params()
    r_feet = np.ascontiguousarray(r_feet, dtype=np.float64).reshape(-1, 3)
    nS = r_feet.shape[0]
    n, m = 3 * nS, 5 * nS
    A, S, b, W = np.zeros((6, n)), np.zeros(6), np.zeros(6), np.zeros(n)
    D, d, f = np.zeros((m, n)), np.zeros(m), np.zeros(m)
    a = [_d(v) for v in (r_feet, wrench, np.reshape(n_B, (nS, 3)), np.reshape(t1, (nS, 3)), np.reshape(t2, (nS, 3)))]
    lib().oracle_force_lsq_assemble(C.byref(prm), nS, a[0][1], a[1][1], a[2][1], a[3][1], a[4][1],
                                    *[v.ctypes.data_as(_dp) for v in (A, S, b, W, D, d, f)])
    return A, S, b, W, D, d, f


def weighted_lsq_qp(A, S, b, W, Ceq=None, ceq=None, D=None, d=None, f=None):
    """oracle_weighted_lsq_qp on one problem -> (x, status)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    k, n = A.shape
    p = 0 if Ceq is None else int(np.asarray(Ceq).shape[0])
    m = 0 if D is None else int(np.asarray(D).shape[0])
    arrs = [_d(v) if v is not None else (None, None) for v in (A, S, b, W, Ceq, ceq, D, d, f)]
    x = np.zeros(n)
    fn = lib().oracle_weighted_lsq_qp
    fn.restype = C.c_int
    st = fn(n, k, p, m, *[a[1] for a in arrs], x.ctypes.data_as(_dp))
    return x, int(st)


# ---- pose optimisation (config 5) ---------------------------------------------------------------
class PoseProblem(C.Structure):
    _fields_ = [("n_legs", C.c_int), ("leg_order", C.c_int * 4), ("stance", (C.c_double * 3) * 4),
                ("nominal", (C.c_double * 3) * 4), ("hips", (C.c_double * 3) * 4), ("max_len", C.c_double * 4),
                ("n_vertices", C.c_int), ("polygon", (C.c_double * 2) * 4), ("r_com", C.c_double * 3),
                ("com_weight", C.c_double)]


def pose_problem(pb, i, hips, leg_order, com_weight=2.0):
    """Problem i of a batch dict (synth.make_pose_problems layout)."""
    p = PoseProblem()
    mask = pb["stance_mask"][i]
    legs = [l for l in leg_order if mask[l]]
    p.n_legs = len(legs)
    for k, l in enumerate(legs):
        p.leg_order[k] = int(l)
    for l in range(4):
        for a in range(3):
            p.stance[l][a] = pb["stance"][i][l][a]
            p.nominal[l][a] = pb["nominal"][i][l][a]
            p.hips[l][a] = hips[l][a]
        p.max_len[l] = pb["max_len"][i][l]
        p.polygon[l][0], p.polygon[l][1] = pb["polygon"][i][l]
    p.n_vertices = int(pb["n_vertices"][i])
    for a in range(3):
        p.r_com[a] = pb["r_com"][i][a]
    p.com_weight = com_weight
    return p


def pose_sqp(pb, i, hips, leg_order, tol=0.05, max_iter=30, dummy_equality=1, com_weight=2.0):
    lib().oracle_pose_cost.restype = C.c_double
    p = pose_problem(pb, i, hips, leg_order, com_weight)
    pose_in = (C.c_double * 7)(*pb["pose"][i])
    out = (C.c_double * 7)()
    it, cost = C.c_int(), C.c_double()
    hist = np.zeros((max(max_iter, 1), 6))
    st = lib().oracle_pose_sqp(C.byref(p), pose_in, C.c_double(tol), int(max_iter), int(dummy_equality), out,
                               C.byref(it), C.byref(cost), hist.ctypes.data_as(_dp))
    return dict(pose=np.array(out[:]), iters=it.value, cost=cost.value, status=st, dp=hist[:it.value].copy())


def pose_cost(pb, i, pose, hips, leg_order, com_weight=2.0):
    lib().oracle_pose_cost.restype = C.c_double
    p = pose_problem(pb, i, hips, leg_order, com_weight)
    return lib().oracle_pose_cost(C.byref(p), (C.c_double * 7)(*pose))


# ---- swing-leg torque (row a18) ---------------------------------------------------------------
class SwingParams(C.Structure):
    _fields_ = [("kp", C.c_double * 3), ("kd", C.c_double * 3), ("period", C.c_double), ("accel_window", C.c_double),
                ("accel_scale", C.c_double), ("gravity", C.c_double)]


def default_swing_params():
    p = SwingParams()
    lib().oracle_swing_default_params(C.byref(p))
    return p


def leg_rnea(leg, q, qd, qdd, g):
    a = [_d(v) for v in (q, qd, qdd, g)]
    tau = np.zeros(3)
    lib().oracle_leg_rnea(int(leg), a[0][1], a[1][1], a[2][1], a[3][1], tau.ctypes.data_as(_dp))
    return tau


def swing_leg_torque(leg, q_id, q, qd, qd_oldest, target_pos, target_vel, params=None):
    prm = params or default_swing_params()
    a = [_d(v) for v in (q_id, q, qd, qd_oldest, target_pos, target_vel)]
    tau = np.zeros(3)
    lib().oracle_swing_leg_torque(C.byref(prm), int(leg), *[x[1] for x in a], tau.ctypes.data_as(_dp))
    return tau


def swing_batch(q, qd, qd_oldest, target_pos, target_vel, support, q_id=None, params=None):
    """[B,12] arrays; returns tau [B,12] (0 for support legs)."""
    B = q.shape[0]
    tau = np.zeros((B, 12))
    qi = q if q_id is None else q_id
    for i in range(B):
        for l in range(4):
            if not support[i][l]:
                sl = slice(3 * l, 3 * l + 3)
                tau[i, sl] = swing_leg_torque(l, qi[i, sl], q[i, sl], qd[i, sl], qd_oldest[i, sl], target_pos[i, sl],
                                              target_vel[i, sl], params)
    return tau


def pose_qp(pb, i, hips, leg_order, dummy_equality=1):
    p = pose_problem(pb, i, hips, leg_order)
    out = (C.c_double * 7)()
    st = lib().oracle_pose_qp(C.byref(p), (C.c_double * 7)(*pb["pose"][i]), int(dummy_equality), out)
    return dict(pose=np.array(out[:]), status=st)


def pose_geometric(pb, i, hips, leg_order, stance_for_orientation=None):
    """PoseOptimizationGeometric::optimize; stance_for_orientation [4][3] by limb id (default: the stance)."""
    p = pose_problem(pb, i, hips, leg_order)
    sfo = np.ascontiguousarray(pb["stance"][i] if stance_for_orientation is None else stance_for_orientation, dtype=np.float64)
    out, qp = (C.c_double * 7)(), (C.c_double * 4)()
    st = lib().oracle_pose_geometric(C.byref(p), sfo.ctypes.data_as(_dp), out, qp)
    return dict(pose=np.array(out[:]), q_procrustes=np.array(qp[:]), status=st)


def base_auto_optimize_pose(pb, i, hips, leg_order, stance_for_orientation=None, min_len=(0.1, 0.1, 0.1, 0.1), leg_tol=0.0,
                            tol=0.05, max_iter=30, dummy_equality=1):
    """BaseAuto::optimizePose: geometric -> QP -> check -> SQP.  Returns dict(pose, stage, status)."""
    p = pose_problem(pb, i, hips, leg_order)
    sfo = np.ascontiguousarray(pb["stance"][i] if stance_for_orientation is None else stance_for_orientation, dtype=np.float64)
    out, stage = (C.c_double * 7)(), C.c_int()
    st = lib().oracle_base_auto_optimize_pose(C.byref(p), sfo.ctypes.data_as(_dp), (C.c_double * 4)(*min_len), C.c_double(leg_tol),
                                              C.c_double(tol), int(max_iter), int(dummy_equality), out, C.byref(stage))
    return dict(pose=np.array(out[:]), stage=stage.value, status=st)


def sym4_eigen(Cm):
    Cm = np.ascontiguousarray(Cm, dtype=np.float64)
    w, V = np.zeros(4), np.zeros((4, 4))
    lib().oracle_sym4_eigen(Cm.ctypes.data_as(_dp), w.ctypes.data_as(_dp), V.ctypes.data_as(_dp))
    return w, V


def pose_check(pb, i, pose, hips, leg_order, min_len=(0.2, 0.2, 0.2, 0.2), leg_tol=0.0):
    p = pose_problem(pb, i, hips, leg_order)
    return int(lib().oracle_pose_check(C.byref(p), (C.c_double * 7)(*pose), (C.c_double * 4)(*min_len), C.c_double(leg_tol)))


def leg_state_machine(io, i, index_quirk=1):
    """oracle_leg_state_machine on robot i of the arrays in `io` (fields of qlamd_leg_state_batch); in/out arrays are
    updated in place."""
    u8, i8 = C.POINTER(C.c_uint8), C.POINTER(C.c_int8)
    row = lambda name, ty: io[name][i:i + 1].ctypes.data_as(ty)  # noqa: E731
    lib().oracle_leg_state_machine(row("support_leg", u8), row("phase", _dp), row("is_footstep", u8), row("contact", u8),
                                   row("joint_position", _dp), int(index_quirk), row("limb_state", i8), row("store_flag", u8),
                                   row("stored_joint_position", _dp), row("joint_command", _dp), row("foot_target", _dp),
                                   row("support", u8), row("leg_state_code", i8))


class RobotStateFields(C.Structure):
    _fields_ = [("des_pos", C.c_double * 3), ("des_quat", C.c_double * 4), ("des_linvel", C.c_double * 3),
                ("des_angvel", C.c_double * 3), ("joint_command", C.c_double * 12), ("foot_position", C.c_double * 12),
                ("foot_velocity", C.c_double * 12), ("foot_acceleration", C.c_double * 12), ("surface_normal", C.c_double * 12),
                ("phase", C.c_double * 4), ("support_leg", C.c_uint8 * 4), ("leg_mode", C.c_uint8 * 4)]

    def as_dict(self):
        return {n: np.array(getattr(self, n)[:]) for n, _ in self._fields_}


def robot_state_unpack(msg):
    """oracle_robot_state_unpack on one serialised free_gait_msgs/RobotState.  Returns (dict, status)."""
    f = RobotStateFields()
    buf = (C.c_uint8 * max(len(msg), 1)).from_buffer_copy(bytes(msg) if len(msg) else b"\0")
    st = lib().oracle_robot_state_unpack(buf, C.c_size_t(len(msg)), C.byref(f))
    return f.as_dict(), st


IK_REFERENCE_GEOMETRY = (0.1, 0.25, 0.25)   # d, l1, l2 hard-coded at quadrupedkinematics.cpp:383-385
IK_DEFAULT_CONFIG = (2, 0, 2, 0)            # "><": LF IN_LEFT, RF OUT_LEFT, RH IN_LEFT, LH OUT_LEFT (quadruped_state.cpp:61,385-390)


def leg_ik(leg, p_base, config, geom=IK_REFERENCE_GEOMETRY):
    p, pp = _d(p_base)
    q = np.zeros(3)
    ok = lib().oracle_leg_ik(int(leg), pp, int(config), (C.c_double * 3)(*geom), q.ctypes.data_as(_dp))
    return q, ok


class PidParams(C.Structure):
    _fields_ = [(n, C.c_double * 12) for n in ("p", "i", "d", "i_max", "i_min")] + [("antiwindup", C.c_int)] + \
               [(n, C.c_double * 12) for n in ("lower", "upper")]


def default_pid_params():
    p = PidParams()
    lib().oracle_pid_default_params(C.byref(p))
    return p


def swing_branch_leg(leg, leg_mode, base_quat, q_id, q, qd, qd_oldest, target_pos, target_vel, joint_command, period,
                     e_last, e_int, params=None, pid=None):
    """oracle_swing_branch_leg; e_last / e_int (3-vectors, float64) are updated in place.  Returns effort[3]."""
    prm = params or default_swing_params()
    pp = pid or default_pid_params()
    a = [_d(v) for v in (base_quat, q_id, q, qd, qd_oldest, target_pos, target_vel, joint_command)]
    eff = np.zeros(3)
    lib().oracle_swing_branch_leg(C.byref(prm), C.byref(pp), int(leg), int(leg_mode), *[x[1] for x in a], C.c_double(period),
                                  e_last.ctypes.data_as(_dp), e_int.ctypes.data_as(_dp), eff.ctypes.data_as(_dp))
    return eff


def pose_sqp_batch(pb, hips, leg_order, tol=0.05, max_iter=30, dummy_equality=1, nthreads=1, problems=None):
    """oracle_pose_sqp_batch over a whole batch dict.  Returns (pose [B,7], iters [B], status [B], problems) --
    pass `problems` back in to skip the marshalling on repeated calls (timing loops)."""
    B = pb["pose"].shape[0]
    if problems is None:
        problems = (PoseProblem * B)()
        for i in range(B):
            problems[i] = pose_problem(pb, i, hips, leg_order)
    pose_in = np.ascontiguousarray(pb["pose"], dtype=np.float64)
    out = np.zeros((B, 7)); it = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
    lib().oracle_pose_sqp_batch(problems, pose_in.ctypes.data_as(_dp), C.c_longlong(B), C.c_double(tol), int(max_iter),
                                int(dummy_equality), out.ctypes.data_as(_dp), it.ctypes.data_as(C.POINTER(C.c_int)),
                                st.ctypes.data_as(C.POINTER(C.c_int)), int(nthreads))
    return out, it, st, problems


# ---- whole-body (floating-base) dynamics and QP: oracle_wholebody.c (parity unpinned) ----
class WbParams(C.Structure):
    _fields_ = [("force_weights", C.c_double * 6), ("regularizer", C.c_double), ("torque_weight", C.c_double),
                ("friction", C.c_double), ("min_normal_force", C.c_double), ("torque_limit", C.c_double),
                ("gravity", C.c_double)]


def default_wb_params():
    p = WbParams()
    lib().oracle_wb_default_params(C.byref(p))
    return p


def wb_mass_matrix(q):
    a, ap = _d(q)
    M = np.zeros((18, 18))
    lib().oracle_wb_mass_matrix(ap, M.ctypes.data_as(_dp))
    return M


def wb_inverse_dynamics(q, base_quat, nu, nudot, gravity=9.81):
    a = [_d(v) for v in (q, base_quat, nu, nudot)]
    out = np.zeros(18)
    lib().oracle_wb_inverse_dynamics(*[x[1] for x in a], C.c_double(gravity), out.ctypes.data_as(_dp))
    return out


def wb_nonlinear_effects(q, base_quat, nu, gravity=9.81):
    a = [_d(v) for v in (q, base_quat, nu)]
    out = np.zeros(18)
    lib().oracle_wb_nonlinear_effects(*[x[1] for x in a], C.c_double(gravity), out.ctypes.data_as(_dp))
    return out


def wb_contact_jacobian(q):
    a, ap = _d(q)
    J = np.zeros((12, 18))
    lib().oracle_wb_contact_jacobian(ap, J.ctypes.data_as(_dp))
    return J


def wb_kinetic_energy(q, nu):
    lib().oracle_wb_kinetic_energy.restype = C.c_double
    a, b = _d(q), _d(nu)
    return lib().oracle_wb_kinetic_energy(a[1], b[1])


def wb_potential_energy(q, base_pos, base_quat, gravity=9.81):
    lib().oracle_wb_potential_energy.restype = C.c_double
    a = [_d(v) for v in (q, base_pos, base_quat)]
    return lib().oracle_wb_potential_energy(*[x[1] for x in a], C.c_double(gravity))


def wb_step_batch(s, params=None, nthreads=1):
    """oracle_wb_step_batch over a state dict with q, qd, base_quat, base_linvel, base_angvel, a_des, stance and
    optionally qdd_des, normals.  Returns (tau [B,12], grf [B,12], status [B])."""
    prm = params or default_wb_params()
    B = s["q"].shape[0]
    keep = [np.ascontiguousarray(s[k], dtype=np.float64) for k in ("q", "qd", "base_quat", "base_linvel", "base_angvel", "a_des")]
    qdd = np.ascontiguousarray(s["qdd_des"], dtype=np.float64) if s.get("qdd_des") is not None else None
    nrm = np.ascontiguousarray(s["normals"], dtype=np.float64) if s.get("normals") is not None else None
    stance = np.ascontiguousarray(s["stance"], dtype=np.uint8)
    tau = np.zeros((B, 12)); grf = np.zeros((B, 12)); st = np.zeros(B, np.int32)
    lib().oracle_wb_step_batch(C.byref(prm), C.c_longlong(B), *[k.ctypes.data_as(_dp) for k in keep],
                               qdd.ctypes.data_as(_dp) if qdd is not None else None,
                               stance.ctypes.data_as(C.POINTER(C.c_uint8)),
                               nrm.ctypes.data_as(_dp) if nrm is not None else None,
                               tau.ctypes.data_as(_dp), grf.ctypes.data_as(_dp), st.ctypes.data_as(C.POINTER(C.c_int)),
                               int(nthreads))
    return tau, grf, st


# ---- the whole control tick (oracle_tick.c) ---------------------------------------------------------------
class TickState(C.Structure):
    _fields_ = [("has_command", C.c_int), ("command", RobotStateFields), ("limb_state", C.c_int8 * 4), ("store_flag", C.c_uint8 * 4),
                ("stored_joint_position", C.c_double * 12), ("leg_mode", C.c_uint8 * 4), ("support", C.c_uint8 * 4),
                ("pid_error_last", C.c_double * 12), ("pid_error_integral", C.c_double * 12), ("joint_effort", C.c_double * 12)]


def new_tick_state():
    """A controller that has not run yet: all zero, every leg a support leg (State's constructor)."""
    s = TickState()
    for l in range(4):
        s.support[l] = 1
    return s


def full_tick(s, msg, q, qd, qd_oldest, base_pos, base_quat, base_linvel, base_angvel, contact, period, index_quirk=1,
              keep_on_failure=0, params=None, swing=None, pid=None):
    """oracle_full_tick on one robot; `s` (TickState) is updated in place.  Returns (status, message_status, leg_state_code)."""
    prm, sp, pp = params or default_params(), swing or default_swing_params(), pid or default_pid_params()
    buf = (C.c_uint8 * max(len(msg), 1)).from_buffer_copy(bytes(msg) if len(msg) else b"\0")
    a = [_d(v) for v in (q, qd, qd_oldest, base_pos, base_quat, base_linvel, base_angvel)]
    con = np.ascontiguousarray(contact, dtype=np.uint8)
    code = np.zeros(4, np.int8)
    mst = C.c_int()
    st = lib().oracle_full_tick(C.byref(prm), C.byref(sp), C.byref(pp), buf, C.c_size_t(len(msg)), *[x[1] for x in a],
                                con.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_double(period), int(index_quirk), int(keep_on_failure),
                                C.byref(s), code.ctypes.data_as(C.POINTER(C.c_int8)), C.byref(mst))
    return st, mst.value, code


# ---- pieces of one SQP iteration (tests/tools/gen_sqp_goldens.py drives the reference's compiled solver with them) ----
def pose_grad_hess(p, pose):
    """(g[6], H[6][6]) of a PoseProblem at pose (oracle_pose_grad_hess)."""
    g, H = np.zeros(6), np.zeros((6, 6))
    lib().oracle_pose_grad_hess(C.byref(p), (C.c_double * 7)(*pose), g.ctypes.data_as(_dp), H.ctypes.data_as(_dp))
    return g, H


def pose_constraints(p, pose):
    """(val[m], vmax[m], A[m][6]) of a PoseProblem at pose (oracle_pose_constraints)."""
    val, vmax, A = np.zeros(8), np.zeros(8), np.zeros((8, 6))
    m = lib().oracle_pose_constraints(C.byref(p), (C.c_double * 7)(*pose), val.ctypes.data_as(_dp), vmax.ctypes.data_as(_dp),
                                      A.ctypes.data_as(_dp))
    return val[:m].copy(), vmax[:m].copy(), A[:m].copy()


def quat_box_plus(q, d):
    out = (C.c_double * 4)()
    lib().oracle_quat_box_plus((C.c_double * 4)(*q), (C.c_double * 3)(*d), out)
    return np.array(out[:])
