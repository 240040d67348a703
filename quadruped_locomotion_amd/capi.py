"""ctypes binding of the C-ABI (include/qlamd.h) -- plumbing only.

The arithmetic lives in the HIP kernels of libqlamd.so.  There is no Python or
CPU fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libqlamd.so")

OK = 0
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_LOADED, ERR_OUT_OF_MEMORY, ERR_BUSY, ERR_NEEDS_RESERVE = -1, -2, -3, -4, -5, -6, -7
STATUS_OK, STATUS_INFEASIBLE, STATUS_NOT_PD, STATUS_MAX_ITER = 0, 1, 2, 3
STATUS_WARM_REJECTED = 6
MEM_DEVICE, MEM_HOST = 0, 1

EXPORTS = (
    "qlamd_balance_default_params", "qlamd_default_robot_model", "qlamd_context_create",
    "qlamd_context_destroy", "qlamd_set_robots_per_wave", "qlamd_balance_solve_batch",
    "qlamd_virtual_wrench_batch", "qlamd_leg_kinematics_batch", "qlamd_strerror", "qlamd_version",
    "qlamd_qp_solve_batch", "qlamd_pose_default_params", "qlamd_pose_sqp_batch",
    "qlamd_force_distribution_batch", "qlamd_swing_default_params", "qlamd_swing_leg_torque_batch",
    "qlamd_pose_qp_batch", "qlamd_pose_check_batch", "qlamd_pose_geometric_batch",
    "qlamd_base_auto_optimize_pose_batch", "qlamd_leg_state_machine_batch", "qlamd_robot_state_unpack_batch",
    "qlamd_ik_default_params", "qlamd_leg_inverse_kinematics_batch",
    "qlamd_joint_pid_default_params", "qlamd_swing_branch_batch",
    "qlamd_wholebody_default_params", "qlamd_wholebody_dynamics_batch", "qlamd_wholebody_solve_batch",
    "qlamd_full_tick_batch", "qlamd_set_option", "qlamd_tick_command_bytes", "qlamd_weighted_lsq_qp_batch",
    "qlamd_reserve", "qlamd_balance_solve_placed_batch", "qlamd_force_distribution_placed_batch",
    "qlamd_placement_from_iterations", "qlamd_place_next_call", "qlamd_get_counter",
)


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


class Placement(C.Structure):
    """qlamd_placement"""
    _fields_ = [("robot_order", C.c_void_p), ("iterations", C.c_void_p), ("prev_iterations", C.c_void_p),
                ("next_robot_order", C.c_void_p), ("policy", C.c_int), ("prev_working_set", C.c_void_p),
                ("working_set", C.c_void_p)]


class RobotModel(C.Structure):
    _fields_ = [
        ("joint_xyz", ((C.c_double * 3) * 4) * 4), ("joint_rpy", ((C.c_double * 3) * 4) * 4),
        ("link_mass", (C.c_double * 4) * 4), ("link_com", ((C.c_double * 3) * 4) * 4),
        ("link_inertia", ((C.c_double * 6) * 4) * 4),
        ("base_mass", C.c_double), ("base_com", C.c_double * 3), ("base_inertia", C.c_double * 6),
    ]


class LegStateBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("support_leg", "phase", "is_footstep", "contact", "joint_position", "limb_state",
                                          "store_flag", "stored_joint_position", "joint_command", "foot_target", "support",
                                          "leg_state_code")]


LEG_STATE_DTYPES = dict(support_leg=np.uint8, phase=np.float64, is_footstep=np.uint8, contact=np.uint8,
                        joint_position=np.float64, limb_state=np.int8, store_flag=np.uint8,
                        stored_joint_position=np.float64, joint_command=np.float64, foot_target=np.float64,
                        support=np.uint8, leg_state_code=np.int8)


ROBOT_STATE_FIELDS = (("des_pos", 3), ("des_quat", 4), ("des_linvel", 3), ("des_angvel", 3), ("joint_command", 12),
                      ("foot_position", 12), ("foot_velocity", 12), ("foot_acceleration", 12), ("surface_normal", 12),
                      ("phase", 4))


class RobotStateFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _ in ROBOT_STATE_FIELDS] + [("support_leg", C.c_void_p), ("leg_mode", C.c_void_p)]


class JointPidParams(C.Structure):
    _fields_ = [(n, C.c_double * 12) for n in ("p", "i", "d", "i_max", "i_min", "lower", "upper")] + [("antiwindup", C.c_int)]


class SwingBranchExtra(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("base_orientation", "joint_command", "leg_mode", "pid_error_last", "pid_error_integral")]


TICK_FIELDS = (("messages", np.uint8), ("offsets", np.int64), ("joint_position", np.float64), ("joint_velocity", np.float64),
               ("joint_velocity_oldest", np.float64), ("base_position", np.float64), ("base_orientation", np.float64),
               ("base_linear_velocity", np.float64), ("base_angular_velocity", np.float64), ("contact", np.uint8),
               ("limb_state", np.int8), ("store_flag", np.uint8), ("stored_joint_position", np.float64), ("leg_mode", np.uint8), ("support", np.uint8),
               ("pid_error_last", np.float64), ("pid_error_integral", np.float64), ("joint_effort", np.float64),
               ("leg_state_code", np.int8), ("status", np.int32), ("message_status", np.int32), ("command", np.uint8),
               ("working_set", np.uint32), ("placement_state", np.int32))


class TickBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _ in TICK_FIELDS]


class WholebodyBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("joint_position", "joint_velocity", "base_orientation", "base_linear_velocity",
                                           "base_angular_velocity", "desired_base_acceleration",
                                           "desired_joint_acceleration", "support_leg", "surface_normal")]


class WholebodyParams(C.Structure):
    _fields_ = [("torque_weight", C.c_double), ("torque_limit", C.c_double), ("gravity", C.c_double)]


# key of a synth.make_wholebody_states dict -> field of qlamd_wholebody_batch
WHOLEBODY_FIELDS = (("q", "joint_position"), ("qd", "joint_velocity"), ("base_quat", "base_orientation"),
                    ("base_linvel", "base_linear_velocity"), ("base_angvel", "base_angular_velocity"),
                    ("a_des", "desired_base_acceleration"), ("qdd_des", "desired_joint_acceleration"),
                    ("stance", "support_leg"), ("normals", "surface_normal"))


class IkParams(C.Structure):
    _fields_ = [("d", C.c_double), ("l1", C.c_double), ("l2", C.c_double), ("limb_config", C.c_uint8 * 4)]


class SwingParams(C.Structure):
    _fields_ = [("kp", C.c_double * 3), ("kd", C.c_double * 3), ("period", C.c_double), ("accel_window", C.c_double),
                ("accel_scale", C.c_double), ("gravity", C.c_double)]


class SwingBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("joint_position", "joint_velocity", "joint_velocity_oldest",
                                           "target_foot_position", "target_foot_velocity", "support_leg",
                                           "id_joint_position")]


class StateBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "joint_position", "base_position", "base_orientation", "base_linear_velocity",
        "base_angular_velocity", "desired_position", "desired_orientation",
        "desired_linear_velocity", "desired_angular_velocity", "support_leg", "surface_normal")]


class PoseParams(C.Structure):
    _fields_ = [("hip_in_base", (C.c_double * 3) * 4), ("com_weight", C.c_double), ("tolerance", C.c_double),
                ("max_iterations", C.c_int), ("dummy_equality", C.c_int), ("leg_order", C.c_int * 4)]


class PoseBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("stance", "stance_mask", "nominal_stance", "support_polygon",
                                           "n_vertices", "center_of_mass", "max_limb_length", "pose")]


# problem-dict key (synth.make_pose_problems) -> PoseBatch field
POSE_FIELD_OF_KEY = (("stance", "stance"), ("stance_mask", "stance_mask"), ("nominal", "nominal_stance"),
                     ("polygon", "support_polygon"), ("n_vertices", "n_vertices"), ("r_com", "center_of_mass"),
                     ("max_len", "max_limb_length"), ("pose", "pose"))


# state-dict key -> StateBatch field (keys as produced by synth.make_states)
FIELD_OF_KEY = (
    ("q", "joint_position", 12), ("base_pos", "base_position", 3), ("base_quat", "base_orientation", 4),
    ("base_linvel", "base_linear_velocity", 3), ("base_angvel", "base_angular_velocity", 3),
    ("des_pos", "desired_position", 3), ("des_quat", "desired_orientation", 4),
    ("des_linvel", "desired_linear_velocity", 3), ("des_angvel", "desired_angular_velocity", 3),
)


class QlamdError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed: %s (%d)" % (what, strerror(code), code))
        self.code = code


_lib = None


def lib():
    """Load libqlamd.so; raise loudly if the HIP extension was not built."""
    global _lib
    if _lib is None:
        # torch (device tensors, streams) carries its own HIP runtime: it has to be the one that initialises, or a later
        # `import torch` in the same process finds no GPU.  Plumbing only -- nothing of torch is used by the library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.qlamd_strerror.restype = C.c_char_p
        L.qlamd_strerror.argtypes = [C.c_int]
        L.qlamd_context_create.argtypes = [C.POINTER(BalanceParams), C.POINTER(RobotModel), C.c_int,
                                           C.POINTER(C.c_void_p)]
        L.qlamd_context_destroy.argtypes = [C.c_void_p]
        L.qlamd_context_destroy.restype = None
        L.qlamd_set_robots_per_wave.argtypes = [C.c_void_p, C.c_int]
        if hasattr(L, "qlamd_get_counter"):  # (absent from builds before 0.6: tools/ab runs older libraries through this module)
            L.qlamd_get_counter.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        if hasattr(L, "qlamd_reserve"):  # (absent from libraries of earlier revisions that tools/experiments/variants.py builds)
            L.qlamd_reserve.argtypes = [C.c_void_p, C.c_int64]
        L.qlamd_balance_solve_batch.argtypes = [C.c_void_p, C.POINTER(StateBatch), C.c_int64, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(L, "qlamd_balance_solve_placed_batch"):
            L.qlamd_balance_solve_placed_batch.argtypes = [C.c_void_p, C.POINTER(StateBatch), C.c_int64, C.POINTER(Placement),
                                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
            L.qlamd_force_distribution_placed_batch.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int64, C.POINTER(Placement)] + [
                C.c_void_p] * 3 + [C.c_int, C.c_void_p]
            L.qlamd_placement_from_iterations.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int,
                                                          C.c_void_p]
            L.qlamd_place_next_call.argtypes = [C.c_void_p, C.POINTER(Placement)]
        L.qlamd_virtual_wrench_batch.argtypes = [C.c_void_p, C.POINTER(StateBatch), C.c_int64, C.c_void_p,
                                                 C.c_int, C.c_void_p]
        L.qlamd_leg_kinematics_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_qp_solve_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [
            C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_force_distribution_batch.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_pose_qp_batch.argtypes = [C.c_void_p, C.POINTER(PoseParams), C.POINTER(PoseBatch), C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_pose_check_batch.argtypes = [C.c_void_p, C.POINTER(PoseParams), C.POINTER(PoseBatch), C.c_void_p, C.c_double,
                                             C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_pose_geometric_batch.argtypes = [C.c_void_p, C.POINTER(PoseParams), C.POINTER(PoseBatch), C.c_void_p, C.c_int64,
                                                 C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_base_auto_optimize_pose_batch.argtypes = [C.c_void_p, C.POINTER(PoseParams), C.POINTER(PoseBatch), C.c_void_p,
                                                          C.c_void_p, C.c_double, C.c_int64, C.c_void_p, C.c_void_p,
                                                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_leg_state_machine_batch.argtypes = [C.c_void_p, C.POINTER(LegStateBatch), C.c_int, C.c_int64, C.c_int, C.c_void_p]
        L.qlamd_robot_state_unpack_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(RobotStateFields),
                                                     C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_ik_default_params.argtypes = [C.POINTER(IkParams)]
        L.qlamd_ik_default_params.restype = None
        L.qlamd_leg_inverse_kinematics_batch.argtypes = [C.c_void_p, C.POINTER(IkParams), C.c_void_p, C.c_void_p, C.c_int64,
                                                         C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_joint_pid_default_params.argtypes = [C.POINTER(JointPidParams)]
        L.qlamd_joint_pid_default_params.restype = None
        L.qlamd_swing_branch_batch.argtypes = [C.c_void_p, C.POINTER(SwingParams), C.POINTER(JointPidParams), C.POINTER(SwingBatch),
                                               C.POINTER(SwingBranchExtra), C.c_double, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_swing_leg_torque_batch.argtypes = [C.c_void_p, C.POINTER(SwingParams), C.POINTER(SwingBatch), C.c_int64,
                                                   C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_pose_sqp_batch.argtypes = [C.c_void_p, C.POINTER(PoseParams), C.POINTER(PoseBatch), C.c_int64,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_full_tick_batch.argtypes = [C.c_void_p, C.POINTER(SwingParams), C.POINTER(JointPidParams), C.POINTER(TickBatch),
                                            C.c_double, C.c_int, C.c_int64, C.c_int, C.c_void_p]
        L.qlamd_wholebody_default_params.argtypes = [C.POINTER(WholebodyParams)]
        L.qlamd_wholebody_default_params.restype = None
        L.qlamd_wholebody_dynamics_batch.argtypes = [C.c_void_p, C.POINTER(WholebodyBatch), C.c_double, C.c_int64, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_wholebody_solve_batch.argtypes = [C.c_void_p, C.POINTER(WholebodyParams), C.POINTER(WholebodyBatch), C.c_int64,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.qlamd_weighted_lsq_qp_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [
            C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def strerror(code):
    return lib().qlamd_strerror(int(code)).decode()


def default_params():
    p = BalanceParams()
    lib().qlamd_balance_default_params(C.byref(p))
    return p


def default_pose_params():
    p = PoseParams()
    lib().qlamd_pose_default_params(C.byref(p))
    return p


def default_robot_model():
    m = RobotModel()
    lib().qlamd_default_robot_model(C.byref(m))
    return m


class Context:
    """RAII wrapper of qlamd_context."""

    def __init__(self, params=None, model=None, device=0):
        self._h = C.c_void_p()
        self.params = params if params is not None else default_params()
        rc = lib().qlamd_context_create(C.byref(self.params), C.byref(model) if model is not None else None,
                                        int(device), C.byref(self._h))
        if rc != OK:
            raise QlamdError(rc, "qlamd_context_create")
        self.device = device

    def close(self):
        if self._h:
            lib().qlamd_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        """qlamd_set_option (OPT_* / ON_FAILURE_* below)."""
        rc = lib().qlamd_set_option(self._h, int(option), int(value))
        if rc != OK:
            raise QlamdError(rc, "qlamd_set_option")

    def counter(self, which):
        """qlamd_get_counter (COUNTER_* below); waits for the device."""
        v = C.c_int64(0)
        if not hasattr(lib(), "qlamd_get_counter"):
            return 0
        rc = lib().qlamd_get_counter(self._h, int(which), C.byref(v))
        if rc != OK:
            raise QlamdError(rc, "qlamd_get_counter")
        return int(v.value)

    def reserve(self, max_batch):
        """Size the context's device scratch for batches up to max_batch (qlamd_reserve): before capturing the whole tick."""
        rc = lib().qlamd_reserve(self._h, C.c_int64(int(max_batch)))
        if rc != OK:
            raise QlamdError(rc, "qlamd_reserve")

    def set_robots_per_wave(self, rpw):
        rc = lib().qlamd_set_robots_per_wave(self._h, int(rpw))
        if rc != OK:
            raise QlamdError(rc, "qlamd_set_robots_per_wave")

    # ---- host (numpy) buffers -------------------------------------------------
    def balance_solve_host(self, state, normals=None, want_forces=True, tau=None, grf=None):
        """tau / grf: C-contiguous float64 [B,12] arrays to write into (what QLAMD_ON_FAILURE_KEEP leaves alone is the
        caller's), fresh zeros otherwise."""
        B = int(np.asarray(state["q"]).reshape(-1, 12).shape[0])
        sb, keep = StateBatch(), []
        for key, field, k in FIELD_OF_KEY:
            a = np.ascontiguousarray(np.asarray(state[key], dtype=np.float64).reshape(B, k))
            keep.append(a)
            setattr(sb, field, a.ctypes.data)
        st = np.ascontiguousarray(np.asarray(state["stance"], dtype=np.uint8).reshape(B, 4))
        sb.support_leg = st.ctypes.data
        if normals is not None:
            nw = np.ascontiguousarray(np.asarray(normals, dtype=np.float64).reshape(B, 12))
            keep.append(nw)
            sb.surface_normal = nw.ctypes.data
        tau = _host_out(tau, B, "tau")
        grf = _host_out(grf, B, "grf") if want_forces else None
        status = np.full(B, -1, dtype=np.int32)
        rc = lib().qlamd_balance_solve_batch(self._h, C.byref(sb), B, tau.ctypes.data,
                                             grf.ctypes.data if want_forces else None, status.ctypes.data,
                                             MEM_HOST, None)
        if rc != OK:
            raise QlamdError(rc, "qlamd_balance_solve_batch")
        return tau, grf, status

    # ---- device (torch) buffers ------------------------------------------------
    def balance_solve_device(self, dstate, tau, grf, status, stream=None):
        """dstate: dict of torch CUDA tensors (keys of synth.make_states, plus optional
        'normals'); tau/grf/status: preallocated CUDA tensors.  Asynchronous."""
        sb = StateBatch()
        B = dstate["q"].shape[0]
        for key, field, _ in FIELD_OF_KEY:
            setattr(sb, field, dstate[key].data_ptr())
        sb.support_leg = dstate["stance"].data_ptr()
        if dstate.get("normals") is not None:
            sb.surface_normal = dstate["normals"].data_ptr()
        rc = lib().qlamd_balance_solve_batch(self._h, C.byref(sb), B, tau.data_ptr(),
                                             grf.data_ptr() if grf is not None else None, status.data_ptr(),
                                             MEM_DEVICE, C.c_void_p(stream) if stream else None)
        if rc != OK:
            raise QlamdError(rc, "qlamd_balance_solve_batch")

    def balance_solve_placed_device(self, dstate, tau, grf, status, order=None, iterations=None, prev_iterations=None,
                                    next_order=None, policy=0, stream=None, prev_working_set=None, working_set=None):
        """qlamd_balance_solve_placed_batch on torch CUDA tensors: order = int32 [B] permutation (slot -> robot) or None,
        iterations = int32 [B] output or None; prev_iterations / next_order = the counts of the previous call and the
        placement for the next one (both or neither).  Asynchronous."""
        sb = StateBatch()
        B = dstate["q"].shape[0]
        for key, field, _ in FIELD_OF_KEY:
            setattr(sb, field, dstate[key].data_ptr())
        sb.support_leg = dstate["stance"].data_ptr()
        if dstate.get("normals") is not None:
            sb.surface_normal = dstate["normals"].data_ptr()
        for name, t in (("order", order), ("iterations", iterations), ("prev_iterations", prev_iterations), ("next_order", next_order)):
            if t is not None and (str(t.dtype) != "torch.int32" or t.numel() != B or not t.is_contiguous()):
                raise ValueError("%s must be a contiguous int32 tensor of %d elements" % (name, B))
        for name, t in (("prev_working_set", prev_working_set), ("working_set", working_set)):
            if t is not None and (str(t.dtype) != "torch.int32" or t.numel() != B or not t.is_contiguous()):
                raise ValueError("%s must be a contiguous int32 tensor of %d elements (the 32 bits of a uint32)" % (name, B))
        pl = Placement(_ptr(order), _ptr(iterations), _ptr(prev_iterations), _ptr(next_order), int(policy),
                       _ptr(prev_working_set), _ptr(working_set))
        rc = lib().qlamd_balance_solve_placed_batch(self._h, C.byref(sb), B, C.byref(pl), tau.data_ptr(),
                                                    grf.data_ptr() if grf is not None else None, status.data_ptr(),
                                                    MEM_DEVICE, C.c_void_p(stream) if stream else None)
        if rc != OK:
            raise QlamdError(rc, "qlamd_balance_solve_placed_batch")

    def balance_solve_placed_host(self, state, order=None, normals=None, want_forces=True, prev_iterations=None, policy=0):
        """qlamd_balance_solve_placed_batch with host (numpy) buffers -> (tau, grf, status, iterations[, next_order])."""
        B = int(np.asarray(state["q"]).reshape(-1, 12).shape[0])
        sb, keep = StateBatch(), []
        for key, field, k in FIELD_OF_KEY:
            a = np.ascontiguousarray(np.asarray(state[key], dtype=np.float64).reshape(B, k))
            keep.append(a)
            setattr(sb, field, a.ctypes.data)
        st = np.ascontiguousarray(np.asarray(state["stance"], dtype=np.uint8).reshape(B, 4))
        sb.support_leg = st.ctypes.data
        if normals is not None:
            nw = np.ascontiguousarray(np.asarray(normals, dtype=np.float64).reshape(B, 12))
            keep.append(nw)
            sb.surface_normal = nw.ctypes.data
        if order is not None:
            order = np.ascontiguousarray(np.asarray(order, dtype=np.int32).reshape(B))
        nxt = None
        if prev_iterations is not None:
            prev_iterations = np.ascontiguousarray(np.asarray(prev_iterations, dtype=np.int32).reshape(B))
            nxt = np.full(B, -1, dtype=np.int32)
        tau = np.zeros((B, 12))
        grf = np.zeros((B, 12)) if want_forces else None
        status = np.full(B, -1, dtype=np.int32)
        iters = np.full(B, -1, dtype=np.int32)
        pl = Placement(_ptr(order), iters.ctypes.data, _ptr(prev_iterations), _ptr(nxt), int(policy))
        rc = lib().qlamd_balance_solve_placed_batch(self._h, C.byref(sb), B, C.byref(pl), tau.ctypes.data,
                                                    grf.ctypes.data if want_forces else None, status.ctypes.data, MEM_HOST, None)
        if rc != OK:
            raise QlamdError(rc, "qlamd_balance_solve_placed_batch")
        return (tau, grf, status, iters) if nxt is None else (tau, grf, status, iters, nxt)

    def place_next_call(self, order=None, iterations=None, prev_iterations=None, next_order=None, policy=0):
        """qlamd_place_next_call with torch int32 CUDA tensors (or all None to withdraw a pending placement)."""
        if order is None and iterations is None and next_order is None:
            rc = lib().qlamd_place_next_call(self._h, None)
        else:
            pl = Placement(_ptr(order), _ptr(iterations), _ptr(prev_iterations), _ptr(next_order), int(policy))
            rc = lib().qlamd_place_next_call(self._h, C.byref(pl))
        if rc != OK:
            raise QlamdError(rc, "qlamd_place_next_call")

    def placement_from_iterations(self, iterations, order=None, policy=0, stream=None):
        """qlamd_placement_from_iterations: numpy int32 [B] -> numpy order (synchronous), or torch int32 CUDA tensors
        (order preallocated, asynchronous on `stream`)."""
        if hasattr(iterations, "data_ptr"):
            B = iterations.numel()
            if order is None or str(order.dtype) != "torch.int32" or order.numel() != B or str(iterations.dtype) != "torch.int32":
                raise ValueError("device memory: pass int32 tensors iterations and order of equal length")
            rc = lib().qlamd_placement_from_iterations(self._h, iterations.data_ptr(), B, int(policy), order.data_ptr(),
                                                       MEM_DEVICE, C.c_void_p(stream) if stream else None)
        else:
            it = np.ascontiguousarray(np.asarray(iterations, dtype=np.int32).reshape(-1))
            B = it.shape[0]
            order = np.full(B, -1, dtype=np.int32)
            rc = lib().qlamd_placement_from_iterations(self._h, it.ctypes.data, B, int(policy), order.ctypes.data, MEM_HOST, None)
        if rc != OK:
            raise QlamdError(rc, "qlamd_placement_from_iterations")
        return order

    def virtual_wrench_device(self, dstate, wrench, stream=None):
        sb = StateBatch()
        for key, field, _ in FIELD_OF_KEY:
            setattr(sb, field, dstate[key].data_ptr())
        sb.support_leg = dstate["stance"].data_ptr()
        rc = lib().qlamd_virtual_wrench_batch(self._h, C.byref(sb), dstate["q"].shape[0], wrench.data_ptr(),
                                              MEM_DEVICE, C.c_void_p(stream) if stream else None)
        if rc != OK:
            raise QlamdError(rc, "qlamd_virtual_wrench_batch")

    def leg_kinematics_device(self, q, quat, foot=None, jac=None, grav=None, stream=None):
        rc = lib().qlamd_leg_kinematics_batch(
            self._h, q.data_ptr(), quat.data_ptr(), q.shape[0],
            foot.data_ptr() if foot is not None else None, jac.data_ptr() if jac is not None else None,
            grav.data_ptr() if grav is not None else None, MEM_DEVICE, C.c_void_p(stream) if stream else None)
        if rc != OK:
            raise QlamdError(rc, "qlamd_leg_kinematics_batch")


def virtual_wrench(ctx, state):
    """qlamd_virtual_wrench_batch on host buffers (state dict as synth.make_states) -> wrench [B,6]."""
    B = int(np.asarray(state["base_pos"]).shape[0])
    sb, keep = StateBatch(), []
    for key, field, k in FIELD_OF_KEY:
        if key == "q":
            continue
        a = np.ascontiguousarray(np.asarray(state[key], dtype=np.float64).reshape(B, k))
        keep.append(a)
        setattr(sb, field, a.ctypes.data)
    w = np.zeros((B, 6))
    rc = lib().qlamd_virtual_wrench_batch(ctx._h, C.byref(sb), B, w.ctypes.data, MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_virtual_wrench_batch")
    return w


def leg_kinematics(ctx, q, quat):
    """qlamd_leg_kinematics_batch on host buffers -> (foot [B,4,3], jacobian [B,4,9], gravity torque [B,4,3])."""
    q = np.ascontiguousarray(q, dtype=np.float64); quat = np.ascontiguousarray(quat, dtype=np.float64)
    B = q.shape[0]
    foot, jac, grav = np.zeros((B, 4, 3)), np.zeros((B, 4, 9)), np.zeros((B, 4, 3))
    rc = lib().qlamd_leg_kinematics_batch(ctx._h, q.ctypes.data, quat.ctypes.data, B, foot.ctypes.data, jac.ctypes.data,
                                          grav.ctypes.data, MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_leg_kinematics_batch")
    return foot, jac, grav


def _ptr(a):
    """data pointer of a numpy array or a torch tensor (None -> NULL)."""
    if a is None:
        return None
    return a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data


def _host_out(a, B, name):
    """A caller-supplied host output array the library writes B * 96 bytes into: float64, C-contiguous, [B, 12] -- or a
    fresh one.  (The library cannot see a numpy array's dtype or strides: a float32 or transposed array would be
    overrun.)"""
    if a is None:
        return np.zeros((B, 12))
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and a.shape == (B, 12)):
        raise ValueError("%s must be a C-contiguous float64 array of shape (%d, 12)" % (name, B))
    return a


def force_distribution(ctx, q, quat, support, wrench, normals=None, memory=MEM_HOST, tau=None, grf=None):
    """qlamd_force_distribution_batch with host (numpy) buffers -> (tau, grf, status)."""
    q = np.ascontiguousarray(q, dtype=np.float64); quat = np.ascontiguousarray(quat, dtype=np.float64)
    support = np.ascontiguousarray(support, dtype=np.uint8); wrench = np.ascontiguousarray(wrench, dtype=np.float64)
    normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float64)
    B = q.shape[0]
    if memory == MEM_HOST:
        tau, grf = _host_out(tau, B, "tau"), _host_out(grf, B, "grf")
    elif tau is None or grf is None:
        raise ValueError("device memory: pass preallocated tau / grf tensors")
    st = np.full(B, -1, dtype=np.int32)
    rc = lib().qlamd_force_distribution_batch(ctx._h, _ptr(q), _ptr(quat), _ptr(support), _ptr(normals), _ptr(wrench), B,
                                              _ptr(tau), _ptr(grf), _ptr(st), memory, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_force_distribution_batch")
    return tau, grf, st


def default_swing_params():
    p = SwingParams()
    lib().qlamd_swing_default_params(C.byref(p))
    return p


def swing_leg_torque(ctx, q, qd, qd_oldest, target_pos, target_vel, support, q_id=None, params=None, memory=MEM_HOST,
                     out=None, stream=None):
    """qlamd_swing_leg_torque_batch; numpy arrays for MEM_HOST (returns tau [B,12]), torch tensors + out for MEM_DEVICE."""
    prm = params if params is not None else default_swing_params()
    arrs = [q, qd, qd_oldest, target_pos, target_vel, support, q_id]
    if memory == MEM_HOST:
        arrs = [None if a is None else np.ascontiguousarray(a) for a in arrs]
    sb = SwingBatch(*[_ptr(a) for a in arrs])
    B = int(arrs[0].shape[0])
    tau = np.zeros((B, 12)) if memory == MEM_HOST else out
    rc = lib().qlamd_swing_leg_torque_batch(ctx._h, C.byref(prm), C.byref(sb), B, _ptr(tau), memory,
                                            C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_swing_leg_torque_batch")
    return tau


def default_joint_pid_params():
    p = JointPidParams()
    lib().qlamd_joint_pid_default_params(C.byref(p))
    return p


def swing_branch(ctx, joint_effort, q, qd, qd_oldest, target_pos, target_vel, support, base_orientation, joint_command,
                 leg_mode, pid_error_last, pid_error_integral, period, q_id=None, params=None, pid=None, memory=MEM_HOST,
                 stream=None):
    """qlamd_swing_branch_batch.  joint_effort, pid_error_last, pid_error_integral are updated in place (host:
    C-contiguous float64 numpy arrays; device: torch CUDA tensors)."""
    prm = params if params is not None else default_swing_params()
    pidp = pid if pid is not None else default_joint_pid_params()
    arrs = [q, qd, qd_oldest, target_pos, target_vel, support, q_id]
    ext = [base_orientation, joint_command, leg_mode, pid_error_last, pid_error_integral]
    if memory == MEM_HOST:
        arrs = [None if a is None else np.ascontiguousarray(a) for a in arrs]
        ext[:3] = [None if a is None else np.ascontiguousarray(a) for a in ext[:3]]
        for a in (joint_effort, pid_error_last, pid_error_integral):
            assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    sb = SwingBatch(*[_ptr(a) for a in arrs])
    ex = SwingBranchExtra(*[_ptr(a) for a in ext])
    B = int(arrs[0].shape[0])
    rc = lib().qlamd_swing_branch_batch(ctx._h, C.byref(prm), C.byref(pidp), C.byref(sb), C.byref(ex), float(period), B,
                                        _ptr(joint_effort), memory, C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_swing_branch_batch")
    return joint_effort


def pose_sqp(ctx, problems, params=None, memory=MEM_HOST, out=None, stream=None):
    """problems: dict as synth.make_pose_problems (numpy for MEM_HOST, torch CUDA tensors for
    MEM_DEVICE).  Returns (pose [B,7], iterations [B], status [B]) -- numpy arrays for host
    memory; for device memory the preallocated tensors passed in `out`."""
    prm = params if params is not None else default_pose_params()
    B = int(problems["pose"].shape[0])
    pb = PoseBatch()
    keep = []
    for key, field in POSE_FIELD_OF_KEY:
        a = problems.get(key)
        if a is not None and memory == MEM_HOST:
            a = np.ascontiguousarray(a)
            keep.append(a)
        setattr(pb, field, _ptr(a))
    if memory == MEM_HOST:
        pose = np.zeros((B, 7)); it = np.zeros(B, dtype=np.int32); st = np.full(B, -1, dtype=np.int32)
    else:
        pose, it, st = out
    rc = lib().qlamd_pose_sqp_batch(ctx._h, C.byref(prm), C.byref(pb), B, _ptr(pose), _ptr(it), _ptr(st), memory,
                                    C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_pose_sqp_batch")
    return pose, it, st


def _pose_batch(problems, memory):
    pb, keep = PoseBatch(), []
    for key, field in POSE_FIELD_OF_KEY:
        a = problems.get(key)
        if a is not None and memory == MEM_HOST:
            a = np.ascontiguousarray(a)
            keep.append(a)
        setattr(pb, field, _ptr(a))
    return pb, keep


def pose_qp(ctx, problems, params=None):
    """qlamd_pose_qp_batch, host buffers -> (pose [B,7], status [B])."""
    prm = params if params is not None else default_pose_params()
    B = int(problems["pose"].shape[0])
    pb, keep = _pose_batch(problems, MEM_HOST)
    pose = np.zeros((B, 7)); st = np.full(B, -1, dtype=np.int32)
    rc = lib().qlamd_pose_qp_batch(ctx._h, C.byref(prm), C.byref(pb), B, _ptr(pose), _ptr(st), MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_pose_qp_batch")
    return pose, st


def pose_check(ctx, problems, min_len=None, leg_tol=0.0, params=None):
    """qlamd_pose_check_batch for the poses in problems['pose'], host buffers -> ok [B] uint8."""
    prm = params if params is not None else default_pose_params()
    B = int(problems["pose"].shape[0])
    pb, keep = _pose_batch(problems, MEM_HOST)
    mn = None if min_len is None else np.ascontiguousarray(min_len, dtype=np.float64)
    ok = np.zeros(B, dtype=np.uint8)
    rc = lib().qlamd_pose_check_batch(ctx._h, C.byref(prm), C.byref(pb), _ptr(mn), float(leg_tol), B, _ptr(ok), MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_pose_check_batch")
    return ok


def pose_geometric(ctx, problems, stance_for_orientation=None, params=None):
    """qlamd_pose_geometric_batch, host buffers -> pose [B,7]."""
    prm = params if params is not None else default_pose_params()
    B = int(problems["stance"].shape[0])
    pb, keep = _pose_batch(problems, MEM_HOST)
    sfo = None if stance_for_orientation is None else np.ascontiguousarray(stance_for_orientation, dtype=np.float64)
    pose = np.zeros((B, 7))
    rc = lib().qlamd_pose_geometric_batch(ctx._h, C.byref(prm), C.byref(pb), _ptr(sfo), B, _ptr(pose), MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_pose_geometric_batch")
    return pose


def base_auto_optimize_pose(ctx, problems, stance_for_orientation=None, min_len=None, leg_tol=0.0, params=None,
                            memory=MEM_HOST, out=None, stream=None):
    """qlamd_base_auto_optimize_pose_batch -> (pose [B,7], stage [B], iterations [B], status [B]).  Host buffers by
    default; for device memory pass torch CUDA tensors in `problems` and preallocated outputs in `out`."""
    prm = params if params is not None else default_pose_params()
    B = int(problems["stance"].shape[0])
    pb, keep = _pose_batch(problems, memory)
    if memory == MEM_HOST:
        sfo = None if stance_for_orientation is None else np.ascontiguousarray(stance_for_orientation, dtype=np.float64)
        mn = None if min_len is None else np.ascontiguousarray(min_len, dtype=np.float64)
        pose = np.zeros((B, 7)); stage = np.zeros(B, np.int32); it = np.zeros(B, np.int32); st = np.full(B, -1, np.int32)
    else:
        sfo, mn = stance_for_orientation, min_len
        pose, stage, it, st = out
    rc = lib().qlamd_base_auto_optimize_pose_batch(ctx._h, C.byref(prm), C.byref(pb), _ptr(sfo), _ptr(mn), float(leg_tol), B,
                                                   _ptr(pose), _ptr(stage), _ptr(it), _ptr(st), memory,
                                                   C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_base_auto_optimize_pose_batch")
    return pose, stage, it, st


def leg_state_machine(ctx, io, index_quirk=1, memory=MEM_HOST, stream=None):
    """qlamd_leg_state_machine_batch.  `io`: dict with the fields of qlamd_leg_state_batch (numpy arrays of the
    dtypes in LEG_STATE_DTYPES for host memory, torch CUDA tensors for device memory); the in/out and out arrays
    are updated in place."""
    b = LegStateBatch()
    for name, dt in LEG_STATE_DTYPES.items():
        a = io[name]
        if memory == MEM_HOST:
            assert a.dtype == dt and a.flags["C_CONTIGUOUS"], name
        setattr(b, name, _ptr(a))
    B = int(io["phase"].shape[0])
    rc = lib().qlamd_leg_state_machine_batch(ctx._h, C.byref(b), int(index_quirk), B, memory,
                                             C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_leg_state_machine_batch")
    return io


def robot_state_unpack(ctx, messages, offsets, want=None):
    """qlamd_robot_state_unpack_batch on host buffers.  messages: bytes / uint8 array, offsets: int64 [B+1].
    Returns (dict of arrays, status [B]); `want` limits the outputs (default: all)."""
    buf = np.frombuffer(messages, dtype=np.uint8) if isinstance(messages, (bytes, bytearray)) else np.ascontiguousarray(messages, np.uint8)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    B = off.shape[0] - 1
    f, out = RobotStateFields(), {}
    for name, w in ROBOT_STATE_FIELDS:
        if want is None or name in want:
            out[name] = np.zeros((B, w))
            setattr(f, name, _ptr(out[name]))
    for name in ("support_leg", "leg_mode"):
        if want is None or name in want:
            out[name] = np.zeros((B, 4), np.uint8)
            setattr(f, name, _ptr(out[name]))
    st = np.full(B, -1, np.int32)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    rc = lib().qlamd_robot_state_unpack_batch(ctx._h, _ptr(buf), _ptr(off), B, C.byref(f), _ptr(st), MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_robot_state_unpack_batch")
    return out, st


def robot_state_unpack_device(ctx, messages, offsets, stream=None):
    """qlamd_robot_state_unpack_batch on device buffers (torch CUDA tensors: uint8 blob, int64 [B+1] offsets) ->
    (dict of CUDA tensors, status tensor).  Asynchronous."""
    import torch
    B = offsets.numel() - 1
    f, out = RobotStateFields(), {}
    for name, w in ROBOT_STATE_FIELDS:
        out[name] = torch.zeros(B, w, dtype=torch.float64, device=messages.device)
        setattr(f, name, out[name].data_ptr())
    for name in ("support_leg", "leg_mode"):
        out[name] = torch.zeros(B, 4, dtype=torch.uint8, device=messages.device)
        setattr(f, name, out[name].data_ptr())
    st = torch.full((B,), -1, dtype=torch.int32, device=messages.device)
    rc = lib().qlamd_robot_state_unpack_batch(ctx._h, messages.data_ptr(), offsets.data_ptr(), B, C.byref(f), st.data_ptr(), MEM_DEVICE,
                                              C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_robot_state_unpack_batch")
    return out, st


def default_ik_params():
    p = IkParams()
    lib().qlamd_ik_default_params(C.byref(p))
    return p


def leg_inverse_kinematics(ctx, foot_position, joint_position_last=None, params=None):
    """qlamd_leg_inverse_kinematics_batch on host buffers -> (q [B,12], ok [B,4])."""
    prm = params if params is not None else default_ik_params()
    foot = np.ascontiguousarray(foot_position, dtype=np.float64)
    last = None if joint_position_last is None else np.ascontiguousarray(joint_position_last, dtype=np.float64)
    B = foot.shape[0]
    q = np.zeros((B, 12)); ok = np.zeros((B, 4), np.uint8)
    rc = lib().qlamd_leg_inverse_kinematics_batch(ctx._h, C.byref(prm), _ptr(foot), _ptr(last), B, _ptr(q), _ptr(ok), MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_leg_inverse_kinematics_batch")
    return q, ok


def qp_solve(ctx, G, g0, CE, ce0, CI, ci0):
    """Batch of dense QPs with host (numpy) buffers: G [B,n,n], g0 [B,n], CE [B,n,p] or None, CI [B,n,m]."""
    G = np.ascontiguousarray(G, dtype=np.float64)
    B, n = G.shape[0], G.shape[1]
    g0 = np.ascontiguousarray(g0, dtype=np.float64)
    p = 0 if CE is None else int(np.asarray(CE).shape[2])
    m = 0 if CI is None else int(np.asarray(CI).shape[2])
    CE = None if p == 0 else np.ascontiguousarray(CE, dtype=np.float64)
    ce0 = None if p == 0 else np.ascontiguousarray(ce0, dtype=np.float64)
    CI = None if m == 0 else np.ascontiguousarray(CI, dtype=np.float64)
    ci0 = None if m == 0 else np.ascontiguousarray(ci0, dtype=np.float64)
    x = np.zeros((B, n)); f = np.zeros(B); st = np.full(B, -1, dtype=np.int32)
    rc = lib().qlamd_qp_solve_batch(ctx._h, n, p, m, _ptr(G), _ptr(g0), _ptr(CE), _ptr(ce0), _ptr(CI), _ptr(ci0), B,
                                    _ptr(x), _ptr(f), _ptr(st), MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_qp_solve_batch")
    return x, f, st


NO_BOUND = 1.7976931348623157e308   # std::numeric_limits<double>::max(): what the reference writes for "no bound"


def weighted_lsq_qp(ctx, A, S, b, W, Ceq=None, ceq=None, D=None, d=None, f=None, memory=MEM_HOST, out=None, stream=None):
    """qlamd_weighted_lsq_qp_batch: min (Ax-b)'S(Ax-b) + x'Wx  s.t. Cx = c, d <= Dx <= f  (the argument list of
    ooqpei::QuadraticProblemFormulation::solve).  A [B,k,n], S [B,k], b [B,k], W [B,n] (diagonals), Ceq [B,p,n],
    ceq [B,p], D [B,m,n], d / f [B,m].  Host: numpy in, (x, status) out; device: torch tensors and out = (x, status)."""
    arrs = [A, S, b, W, Ceq, ceq, D, d, f]
    if memory == MEM_HOST:
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
    A = arrs[0]
    B, k, n = int(A.shape[0]), int(A.shape[1]), int(A.shape[2])
    p = 0 if arrs[4] is None else int(arrs[4].shape[1])
    m = 0 if arrs[6] is None else int(arrs[6].shape[1])
    if memory == MEM_HOST:
        x, st = np.zeros((B, n)), np.full(B, -1, dtype=np.int32)
    else:
        if out is None:
            raise ValueError("device memory: pass out=(x [B,n] float64, status [B] int32) as preallocated CUDA tensors")
        x, st = out
    rc = lib().qlamd_weighted_lsq_qp_batch(ctx._h, n, k, p, m, *[_ptr(a) for a in arrs], B, _ptr(x), _ptr(st), memory,
                                           C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_weighted_lsq_qp_batch")
    return x, st


def to_device(state, device="cuda:0"):
    """numpy state dict -> dict of torch tensors resident in HBM."""
    import torch
    out = {}
    for k, v in state.items():
        out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device)
    return out


def to_device_records(state, device="cuda:0"):
    """numpy state dict -> dict of torch tensors for a context with OPT_STATE_LAYOUT = STATE_RECORDS: the nine double fields are
    views into ONE [B][48] tensor of qlamd_state_record's (data_ptr() of a view = the field's first element of robot 0), the
    support flags (and normals, if any) arrays of their own."""
    import torch
    B = state["q"].shape[0]
    rec = np.zeros((B, STATE_RECORD_DOUBLES))
    for k, o in STATE_RECORD_OFFSETS.items():
        rec[:, o:o + state[k].shape[1]] = state[k]
    t = torch.from_numpy(rec).to(device)
    out = {k: t[:, o:] for k, o in STATE_RECORD_OFFSETS.items()}
    out["_records"] = t
    for k, v in state.items():
        if k not in STATE_RECORD_OFFSETS:
            out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device)
    return out


def default_wholebody_params():
    p = WholebodyParams()
    lib().qlamd_wholebody_default_params(C.byref(p))
    return p


def _wholebody_batch(state, keep):
    """qlamd_wholebody_batch over numpy arrays (kept alive in `keep`) or torch CUDA tensors."""
    wb = WholebodyBatch()
    for key, field in WHOLEBODY_FIELDS:
        v = state.get(key)
        if v is None:
            continue
        if not hasattr(v, "data_ptr"):
            v = np.ascontiguousarray(v, dtype=np.uint8 if key == "stance" else np.float64)
            keep.append(v)
        setattr(wb, field, _ptr(v))
    return wb


def wholebody_dynamics(ctx, state, gravity=9.81, want=("M", "h", "Jc")):
    """qlamd_wholebody_dynamics_batch on host buffers -> dict with M [B,18,18], h [B,18], Jc [B,12,18]."""
    keep = []
    wb = _wholebody_batch(state, keep)
    B = state["q"].shape[0]
    out = dict(M=np.zeros((B, 18, 18)) if "M" in want else None, h=np.zeros((B, 18)) if "h" in want else None,
               Jc=np.zeros((B, 12, 18)) if "Jc" in want else None)
    rc = lib().qlamd_wholebody_dynamics_batch(ctx._h, C.byref(wb), C.c_double(gravity), B, _ptr(out["M"]), _ptr(out["h"]),
                                              _ptr(out["Jc"]), MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_wholebody_dynamics_batch")
    return out


def wholebody_solve(ctx, state, params=None, tau=None, grf=None):
    """qlamd_wholebody_solve_batch on host buffers -> (tau [B,12], grf [B,12], status [B])."""
    prm = params if params is not None else default_wholebody_params()
    keep = []
    wb = _wholebody_batch(state, keep)
    B = state["q"].shape[0]
    tau, grf = _host_out(tau, B, "tau"), _host_out(grf, B, "grf")
    st = np.full(B, -1, np.int32)
    rc = lib().qlamd_wholebody_solve_batch(ctx._h, C.byref(prm), C.byref(wb), B, _ptr(tau), _ptr(grf), _ptr(st), MEM_HOST, None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_wholebody_solve_batch")
    return tau, grf, st


def wholebody_solve_device(ctx, dstate, tau, grf, status, params=None, stream=None):
    """Same entry on torch CUDA tensors (dstate: dict from to_device); asynchronous."""
    prm = params if params is not None else default_wholebody_params()
    wb = _wholebody_batch(dstate, [])
    rc = lib().qlamd_wholebody_solve_batch(ctx._h, C.byref(prm), C.byref(wb), dstate["q"].shape[0], tau.data_ptr(),
                                           grf.data_ptr() if grf is not None else None, status.data_ptr(), MEM_DEVICE,
                                           C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_wholebody_solve_batch")


def wholebody_dynamics_device(ctx, dstate, M, h, Jc, gravity=9.81, stream=None):
    wb = _wholebody_batch(dstate, [])
    rc = lib().qlamd_wholebody_dynamics_batch(ctx._h, C.byref(wb), C.c_double(gravity), dstate["q"].shape[0],
                                              M.data_ptr() if M is not None else None, h.data_ptr() if h is not None else None,
                                              Jc.data_ptr() if Jc is not None else None, MEM_DEVICE,
                                              C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_wholebody_dynamics_batch")


OPT_ON_FAILURE, OPT_REFINE_PASSES, OPT_DYNAMICS_FORM, OPT_PLACEMENT_WAIT, OPT_WARM_FALLBACK, OPT_STATE_LAYOUT = 1, 2, 5, 6, 7, 8
STATE_FIELDS, STATE_RECORDS, STATE_RECORD_DOUBLES = 0, 1, 48
# offsets (doubles) of the fields inside a qlamd_state_record (include/qlamd.h), by the keys of synth.make_states
STATE_RECORD_OFFSETS = {"q": 0, "base_pos": 12, "base_quat": 16, "base_linvel": 20, "base_angvel": 23, "des_pos": 26, "des_quat": 30,
                        "des_linvel": 34, "des_angvel": 37}
COUNTER_PLACEMENT_GIVE_UPS, COUNTER_WARM_RETRIES = 0, 1
DYNAMICS_AUTO, DYNAMICS_LEG, DYNAMICS_ROW = 0, 1, 2
ON_FAILURE_ZERO, ON_FAILURE_KEEP = 0, 1
PLACEMENT_AUTO, PLACEMENT_LATENCY, PLACEMENT_THROUGHPUT, PLACEMENT_NONE = 0, 1, 2, 3
STATUS_NO_COMMAND = 4
STATUS_DEPENDENT_EQUALITY = 5


def tick_command_bytes(batch):
    """Size of the opaque `command` block of qlamd_tick_batch (zero-filled before the first tick)."""
    fn = lib().qlamd_tick_command_bytes
    fn.restype = C.c_size_t
    fn.argtypes = [C.c_int64]
    return int(fn(int(batch)))


def full_tick(ctx, io, period, index_quirk=1, params=None, pid=None, memory=MEM_HOST, stream=None):
    """qlamd_full_tick_batch.  `io`: dict with the fields of qlamd_tick_batch (TICK_FIELDS: C-contiguous numpy arrays of
    those dtypes for host memory, torch CUDA tensors for device memory; `leg_state_code` and `command` may be None);
    in/out and out arrays are updated in place."""
    prm = params if params is not None else default_swing_params()
    pidp = pid if pid is not None else default_joint_pid_params()
    tb = TickBatch()
    for name, dt in TICK_FIELDS:
        a = io.get(name)
        if a is None:
            continue
        if memory == MEM_HOST:
            assert a.dtype == dt and a.flags["C_CONTIGUOUS"], name
        setattr(tb, name, _ptr(a))
    B = int(io["offsets"].shape[0]) - 1
    rc = lib().qlamd_full_tick_batch(ctx._h, C.byref(prm), C.byref(pidp), C.byref(tb), float(period), int(index_quirk), B, memory,
                                     C.c_void_p(stream) if stream else None)
    if rc != OK:
        raise QlamdError(rc, "qlamd_full_tick_batch")
    return io
