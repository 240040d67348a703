"""Seeded synthetic robot states for the balance-controller hot path.

Distributions follow SURVEY.md section 8(d).  The generator is counter based
(numpy Philox, key = seed): robot i draws the same numbers whatever the batch
size, so a shard [lo, hi) of a larger batch is reproducible on any rank.

Fields are laid out [B][k] exactly as the C-ABI (include/qlamd.h) takes them,
mirroring hardware_interface::RobotStateHandle::Data of the reference
(balance_controller/include/balance_controller/ros_controler/robot_state_interface.hpp:28-65).
"""
import numpy as np

SEED = 20261002
_DRAWS = 64  # uniform draws reserved per robot (fixed so shards line up)

# trot timing, free_gait_ros/test/action_server_test.cpp:183 and
# free_gait_ros/test/gait_generate_client.cpp:82-117 (diagonal pairs)
T_SWING = 0.45
T_STANCE = 0.45
DOUBLE_SUPPORT_FRACTION = 0.10


def _uniform(seed, lo, hi):
    """[hi-lo, _DRAWS] uniforms in [0,1), robot-index addressed."""
    bg = np.random.Philox(key=seed)
    # each Philox counter step yields 4 x 64-bit -> 4 doubles; _DRAWS doubles per robot
    bg.advance(lo * (_DRAWS // 4))
    return np.random.Generator(bg).random((hi - lo, _DRAWS))


def _quat_mul(a, b):
    w1, x1, y1, z1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    w2, x2, y2, z2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                     w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=-1)


def _quat_axis(axis, ang):
    q = np.zeros(ang.shape + (4,))
    q[..., 0] = np.cos(0.5 * ang)
    q[..., 1 + axis] = np.sin(0.5 * ang)
    return q


def _quat_exp(rv):
    ang = np.linalg.norm(rv, axis=-1)
    k = np.where(ang > 1e-12, np.sin(0.5 * ang) / np.maximum(ang, 1e-300), 0.5)
    return np.concatenate([np.cos(0.5 * ang)[..., None], rv * k[..., None]], axis=-1)


# Half-widths of the uniform tracking errors (position m, rotation vector rad, twist m/s or rad/s).
# "calm": a robot holding its pose -- SURVEY.md 8(d) expects "mostly inactive constraints" for the static configs,
# and with the reference gains (Kd = 5000 N s/m) that needs errors this small.  "survey": the literal half-widths
# of SURVEY.md 8(d), which saturate the friction pyramid on most robots (about twice the work per control step).
TRACKING_ERRORS = {"calm": (0.004, 0.005, 0.01), "survey": (0.02, 0.05, 0.1)}


def tracking_error(gait, errors=None):
    """The half-widths make_states uses: static defaults to "calm", trot always takes the survey's."""
    if gait != "static":
        return TRACKING_ERRORS["survey"]
    return TRACKING_ERRORS[errors or "calm"]


_TROT_PHASE_DRAW = 36  # the draw of make_states that is a trot robot's gait phase (12 + 3 + 3 + 3 + 3 + 3 + 3 + 3 + 3)


def trot_stance(phase):
    """Support flags [B][4] (LF, RF, RH, LH) of a trot at gait phase `phase` in [0, 1): one cycle = swing + stance
    (T_SWING + T_STANCE = 0.9 s); pair A = {LF, RH} (legs 0, 2) supports in the first half, pair B = {RF, LH} (legs 1, 3) in
    the second (diagonal pairs, free_gait_ros/test/gait_generate_client.cpp:82-117), and both around the three switches of a
    cycle (phase 0 = 1 and 0.5), a double-support window DOUBLE_SUPPORT_FRACTION wide in all."""
    phase = np.asarray(phase) % 1.0
    half = DOUBLE_SUPPORT_FRACTION / 2.0
    dist_to_switch = np.minimum(np.minimum(phase, np.abs(phase - 0.5)), 1.0 - phase)
    double_support = dist_to_switch < half / 2.0 * 2.0
    a_stance = phase < 0.5
    stance = np.ones(phase.shape + (4,), dtype=np.uint8)
    stance[..., 0] = stance[..., 2] = (a_stance | double_support)
    stance[..., 1] = stance[..., 3] = (~a_stance | double_support)
    return stance


def trot_phase(batch, seed=SEED, offset=0):
    """The gait phase make_states(batch, "trot", seed, offset) drew for each robot."""
    return _uniform(seed, offset, offset + batch)[:, _TROT_PHASE_DRAW]


def make_states(batch, gait="static", seed=SEED, offset=0, errors=None):
    """Return a dict of numpy arrays for robots [offset, offset+batch).

    gait = "static": all four feet in stance (BASELINE configs 1-2).
    gait = "trot":   diagonal-pair trot with a double-support window and a
                     horizontal velocity error sized to load the friction
                     pyramid (BASELINE configs 3-4).
    """
    u = _uniform(seed, offset, offset + batch)
    B = batch
    # static stance: a robot holding its pose (small tracking errors, QP constraints mostly
    # inactive); trot: large tracking errors that load the friction pyramid (active-set churn)
    e_pos, e_rot, e_twist = tracking_error(gait, errors)
    sym = lambda c, half: (2.0 * u[:, c] - 1.0) * half  # noqa: E731
    c = 0

    q_nom = np.tile(np.array([0.0, 0.75, -1.5]), 4)
    q = q_nom[None, :] + np.stack([sym(c + k, 0.15) for k in range(12)], axis=1)
    c += 12
    base_pos = np.array([0.0, 0.0, 0.46])[None, :] + np.stack([sym(c + k, 0.02) for k in range(3)], axis=1)
    c += 3
    yaw, roll, pitch = sym(c, np.pi), sym(c + 1, 0.1), sym(c + 2, 0.1)
    c += 3
    base_quat = _quat_mul(_quat_mul(_quat_axis(2, yaw), _quat_axis(1, pitch)), _quat_axis(0, roll))
    base_linvel = np.stack([sym(c + k, e_twist) for k in range(3)], axis=1)
    c += 3
    base_angvel = np.stack([sym(c + k, e_twist) for k in range(3)], axis=1)
    c += 3
    des_pos = base_pos + np.stack([sym(c + k, e_pos) for k in range(3)], axis=1)
    c += 3
    des_quat = _quat_mul(_quat_exp(np.stack([sym(c + k, e_rot) for k in range(3)], axis=1)), base_quat)
    c += 3
    des_linvel = np.stack([sym(c + k, e_twist) for k in range(3)], axis=1)
    c += 3
    des_angvel = np.stack([sym(c + k, e_twist) for k in range(3)], axis=1)
    c += 3
    assert c == _TROT_PHASE_DRAW
    stance = np.ones((B, 4), dtype=np.uint8)

    if gait == "trot":
        phase = u[:, c]
        ratio = 0.2 + 0.7 * u[:, c + 1]
        theta = 2.0 * np.pi * u[:, c + 2]
        c += 3
        stance = trot_stance(phase)
        # horizontal force demand |F_xy| / F_z ~ U(0.2, 0.9) through the velocity error
        # (F_xy ~ kd * e_v, F_z ~ 51 kg * 9.8), zero horizontal position error
        weight = (27.0 + 4 * 6.0) * 9.8
        ev = ratio * weight / 5000.0
        des_pos[:, 0:2] = base_pos[:, 0:2]
        des_linvel[:, 0] = base_linvel[:, 0] + ev * np.cos(theta)
        des_linvel[:, 1] = base_linvel[:, 1] + ev * np.sin(theta)
    elif gait != "static":
        raise ValueError("gait must be 'static' or 'trot'")

    return dict(q=np.ascontiguousarray(q), base_pos=np.ascontiguousarray(base_pos),
                base_quat=np.ascontiguousarray(base_quat), base_linvel=base_linvel,
                base_angvel=base_angvel, des_pos=np.ascontiguousarray(des_pos),
                des_quat=np.ascontiguousarray(des_quat), des_linvel=np.ascontiguousarray(des_linvel),
                des_angvel=des_angvel, stance=stance)


def _quat_rotate(q, v):
    """v rotated by the unit quaternion q (w, x, y, z), row-wise."""
    w, u = q[:, :1], q[:, 1:]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def next_tick_states(state, dt):
    """The same robots one control period later, to first order in dt: the measured pose integrated with the measured
    twist, the desired pose with the desired twist (base_angvel / des_angvel are expressed in the base frame,
    VirtualModelController.cpp:150-151), joints, twists and stance flags unchanged.  For benchmarks whose placement hints
    must come from OTHER states than the ones being solved (bench.py, tools/experiments/placement_model.py)."""
    s = {k: np.array(v, copy=True) for k, v in state.items()}
    s["base_pos"] = state["base_pos"] + dt * state["base_linvel"]
    s["des_pos"] = state["des_pos"] + dt * state["des_linvel"]
    for pose, twist in (("base_quat", "base_angvel"), ("des_quat", "des_angvel")):
        omega_world = _quat_rotate(state["base_quat"], state[twist])
        s[pose] = _quat_mul(_quat_exp(dt * omega_world), state[pose])
    return s


CONTROL_PERIOD = 0.0025  # 400 Hz, balance_controller/src/ros_controller/balance_controller_manager.cpp:48


def trajectory(batch, gait="static", ticks=1, dt=CONTROL_PERIOD, seed=SEED, offset=0, errors=None):
    """`ticks` consecutive control ticks of the same robots, a list of state dicts: tick 0 = make_states(...), tick t + 1 from
    tick t by integrating the measured pose with the measured twist and the desired pose with the desired twist
    (next_tick_states) AND, for a trot, advancing every robot's gait phase by dt / (T_SWING + T_STANCE) and taking its support
    flags from the new phase (free_gait_ros/test/gait_generate_client.cpp:82-117, action_server_test.cpp:183: 0.45 s of swing,
    0.45 s of stance) -- so a trot batch steps through its contact switches: per tick 4 dt / 0.9 s = 1.1 % of the robots enter
    or leave a double-support window, i.e. change their support set (2 <-> 4 contacts), which is what makes a working set or a
    placement hint of the tick before stale.  Joint angles and twists stay as drawn (the tracking errors drift with the twists'
    difference).  What every benchmark and test of the placed / warm-started loop runs on: hints always come from the states
    of earlier ticks, never from the states being solved."""
    s = make_states(batch, gait, seed, offset, errors)
    out = [s]
    phase = trot_phase(batch, seed, offset) if gait == "trot" else None
    for t in range(1, ticks):
        s = next_tick_states(s, dt)
        if gait == "trot":
            s["stance"] = trot_stance(phase + t * dt / (T_SWING + T_STANCE))
        out.append(s)
    return out


def support_switches(states):
    """Fraction of the robots whose support set differs from the previous tick's, per tick of a trajectory (ticks - 1 numbers)."""
    return [float((a["stance"] != b["stance"]).any(axis=1).mean()) for a, b in zip(states[:-1], states[1:])]


# ---------------------------------------------------------------------------------------------
# Pose-optimisation problems (BASELINE config 5), SURVEY.md section 8(d):
# the SquareUp family of free_gait_core/test/PoseOptimizationSQPTest.cpp:111-199 with seeded
# perturbations.  Limb ids LF, RF, RH, LH = 0..3.
POSE_NOMINAL = np.array([[0.3, 0.2, -0.4], [0.3, -0.2, -0.4], [-0.3, -0.2, -0.4], [-0.3, 0.2, -0.4]])
POSE_FEET = np.array([[0.3, 0.2, -0.1], [0.3, -0.2, -0.1], [-0.3, -0.2, -0.1], [-0.3, 0.2, -0.1]])
POSE_HIPS = np.array([[0.42, 0.075, 0.0], [0.42, -0.075, 0.0], [-0.42, -0.075, 0.0], [-0.42, 0.075, 0.0]])
POSE_MAX_LEN = 0.565          # PoseOptimizationSQPTest.cpp:141
# iteration order of the reference's unordered_map `Stance` when filled LF, RF, LH, RH
# (libstdc++: reverse insertion order), SURVEY.md Q6
POSE_LEG_ORDER = (2, 3, 1, 0)
# counter-clockwise footprint LF, LH, RH, RF (getFootholdsCounterClockwiseOrdered)
_CCW = (0, 3, 2, 1)


def make_pose_problems(batch, seed=SEED + 5, offset=0):
    u = _uniform(seed, offset, offset + batch)
    B = batch
    yaw = np.deg2rad(40.0) * u[:, 0]
    trans = (2.0 * u[:, 1:3] - 1.0) * 0.3
    noise = (2.0 * u[:, 3:15].reshape(B, 4, 3) - 1.0) * 0.05
    c, s = np.cos(yaw), np.sin(yaw)
    Rz = np.zeros((B, 3, 3))
    Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = c, -s, s, c, 1.0
    stance = np.einsum("bij,lj->bli", Rz, POSE_FEET) + noise
    stance[:, :, 0:2] += trans[:, None, :]
    polygon = stance[:, _CCW, 0:2]
    pose = np.zeros((B, 7))
    pose[:, 0:2] = trans
    pose[:, 2] = 0.3
    pose[:, 3] = 1.0
    return dict(stance=np.ascontiguousarray(stance), stance_mask=np.ones((B, 4), dtype=np.uint8),
                nominal=np.ascontiguousarray(np.tile(POSE_NOMINAL, (B, 1, 1))),
                polygon=np.ascontiguousarray(polygon), n_vertices=np.full(B, 4, dtype=np.int32),
                r_com=np.zeros((B, 3)), max_len=np.full((B, 4), POSE_MAX_LEN), pose=pose)


def make_swing_inputs(batch, seed=SEED + 9, offset=0):
    """Swing-leg inputs on top of a trot state batch: newest / oldest joint velocities of the 11-deep
    queue and Cartesian foot targets near the current foot position (row a18)."""
    st = make_states(batch, "trot", offset=offset)
    u = _uniform(seed, offset, offset + batch)
    qd = (2.0 * u[:, 0:12] - 1.0) * 1.5
    qd_old = qd + (2.0 * u[:, 12:24] - 1.0) * 0.2
    dpos = (2.0 * u[:, 24:36] - 1.0) * 0.03
    tvel = (2.0 * u[:, 36:48] - 1.0) * 0.5
    return dict(q=st["q"], qd=np.ascontiguousarray(qd), qd_old=np.ascontiguousarray(qd_old), dpos=np.ascontiguousarray(dpos),
                tvel=np.ascontiguousarray(tvel), support=st["stance"])


# ---------------------------------------------------------------------------------------------
# Whole-body (floating-base) states, SURVEY.md section 8 row f4: the control-step states above plus joint
# velocities and a desired base acceleration [linear ; angular] in base coordinates (what a base-motion
# controller would ask for); the desired joint accelerations are zero (stance legs hold, swing legs coast).
def make_wholebody_states(batch, gait="trot", seed=SEED, offset=0):
    s = make_states(batch, gait, seed, offset)
    u = _uniform(seed + 17, offset, offset + batch)
    sym = lambda c, half: (2.0 * u[:, c] - 1.0) * half  # noqa: E731
    s["qd"] = np.ascontiguousarray(np.stack([sym(k, 0.5) for k in range(12)], axis=1))
    lin = np.stack([sym(12 + k, 1.0) for k in range(3)], axis=1)
    ang = np.stack([sym(15 + k, 2.0) for k in range(3)], axis=1)
    if gait == "trot":  # a horizontal push that loads the friction pyramid, like the velocity error of the trot states
        ratio, theta = 0.2 + 0.7 * u[:, 18], 2.0 * np.pi * u[:, 19]
        lin[:, 0] = 9.81 * ratio * np.cos(theta)
        lin[:, 1] = 9.81 * ratio * np.sin(theta)
    s["a_des"] = np.ascontiguousarray(np.concatenate([lin, ang], axis=1))
    return s


# ---------------------------------------------------------------------------------------------
# Serialised /desired_robot_state messages, one per robot: the command side of a tick.
def make_messages(batch, ragged=False, seed=SEED + 21, mode_names=None):
    """(blob uint8, offsets int64[B+1], fields) -- `fields` holds what the messages carry, [B][k] arrays.
    ragged = False: one publisher's layout (every message has the same field offsets, payloads differ);
    ragged = True: every message has its own string lengths and array counts."""
    from . import wire
    rng = np.random.default_rng(seed)
    names = list(mode_names) if mode_names is not None else list(wire.MODE_NAMES) + ["", "LF_LEG", "footsteps", "Joint"]
    fields = dict(des_pos=rng.normal(size=(batch, 3)), des_quat=rng.normal(size=(batch, 4)), des_linvel=rng.normal(size=(batch, 3)),
                  des_angvel=rng.normal(size=(batch, 3)), joint_command=rng.normal(size=(batch, 12)),
                  foot_position=rng.normal(size=(batch, 12)), foot_velocity=rng.normal(size=(batch, 12)),
                  foot_acceleration=rng.normal(size=(batch, 12)), surface_normal=rng.normal(size=(batch, 12)),
                  phase=rng.random((batch, 4)), support_leg=rng.integers(0, 2, (batch, 4)).astype(np.uint8))
    pick = rng.integers(0, len(names), (batch, 4))
    if not ragged:  # one publisher, one layout: the mode names (strings) are the same for every robot
        pick[:] = pick[0]
    fields["leg_mode"] = np.array([[wire.MODE_CODE.get(names[k], 0) for k in row] for row in pick], dtype=np.uint8)
    one = wire.random_layout(np.random.default_rng(seed + 1))
    msgs = []
    for b in range(batch):
        f = {k: v[b] for k, v in fields.items() if k != "leg_mode"}
        f["mode_name"] = [names[k] for k in pick[b]]
        msgs.append(wire.pack_robot_state(f, wire.random_layout(rng) if ragged else one))
    blob, off = wire.pack_batch(msgs)
    return blob, off, fields


class MessageTemplate:
    """One publisher's layout of free_gait_msgs/RobotState, for packing a whole batch at once: the payload elements of a message sit
    at fixed byte offsets when every message shares the strings and array counts (wire.pack_robot_state's `layout`), so a batch is
    the template's bytes tiled B times with the payload written in place (numpy, no per-robot Python).  The offsets are FOUND, not
    derived: the template message is packed once more per payload element with that element changed, and the bytes that differ
    are where it lives -- whatever wire.pack_robot_state does (the quaternion's x, y, z, w order included) is what this follows."""
    DOUBLES = (("des_pos", 3), ("des_quat", 4), ("des_linvel", 3), ("des_angvel", 3), ("joint_command", 12), ("foot_position", 12),
               ("foot_velocity", 12), ("foot_acceleration", 12), ("surface_normal", 12), ("phase", 4))

    def __init__(self, mode_names, layout=None, seed=SEED + 22):
        from . import wire
        self.layout = layout if layout is not None else wire.random_layout(np.random.default_rng(seed))
        self.mode_names = list(mode_names)
        base = {k: np.arange(1.0, n + 1.0) * (3.0 + i) for i, (k, n) in enumerate(self.DOUBLES)}
        base["support_leg"] = np.zeros(4, np.uint8)
        base["mode_name"] = self.mode_names
        self.template = np.frombuffer(wire.pack_robot_state(base, self.layout), np.uint8).copy()
        self.offsets = {}
        for k, n in self.DOUBLES:
            offs = []
            for e in range(n):
                f = dict(base)
                v = np.array(base[k], copy=True)
                v[e] = -v[e] - 0.5
                f[k] = v
                m = np.frombuffer(wire.pack_robot_state(f, self.layout), np.uint8)
                d = np.nonzero(m != self.template)[0]
                assert len(m) == len(self.template) and len(d) and d.max() - d.min() < 8
                # the element's 8 bytes: the differing bytes lie inside one aligned-to-the-element window; find its start by
                # decoding candidates
                start = None
                for s0 in range(max(0, d.max() - 7), d.min() + 1):
                    if np.frombuffer(m[s0:s0 + 8].tobytes(), "<f8")[0] == v[e]:
                        start = s0
                        break
                assert start is not None
                offs.append(start)
            self.offsets[k] = np.array(offs)
        offs = []
        for e in range(4):
            f = dict(base)
            v = np.zeros(4, np.uint8)
            v[e] = 1
            f["support_leg"] = v
            m = np.frombuffer(wire.pack_robot_state(f, self.layout), np.uint8)
            d = np.nonzero(m != self.template)[0]
            assert len(d) == 1
            offs.append(int(d[0]))
        self.offsets["support_leg"] = np.array(offs)

    def pack(self, fields):
        """(blob uint8 [B * L], offsets int64 [B + 1]) for the [B][k] arrays of `fields` (the keys of DOUBLES + support_leg)."""
        B, L = np.asarray(fields["des_pos"]).shape[0], len(self.template)
        blob = np.tile(self.template, (B, 1))
        for k, n in self.DOUBLES:
            v = np.ascontiguousarray(np.asarray(fields[k], dtype="<f8").reshape(B, n))
            raw = v.view(np.uint8).reshape(B, n, 8)
            for e in range(n):
                o = int(self.offsets[k][e])
                blob[:, o:o + 8] = raw[:, e, :]
        sup = np.asarray(fields["support_leg"]).reshape(B, 4)
        for e in range(4):
            blob[:, int(self.offsets["support_leg"][e])] = (sup[:, e] != 0).astype(np.uint8)
        return np.ascontiguousarray(blob.reshape(-1)), np.arange(B + 1, dtype=np.int64) * L


def wholebody_trajectory(batch, gait="trot", ticks=1, dt=CONTROL_PERIOD, seed=SEED, offset=0):
    """trajectory() for the whole-body step: the base as there, the joints moved by their own rates, rates and the desired base
    acceleration as drawn (make_wholebody_states)."""
    first = make_wholebody_states(batch, gait, seed, offset)
    base = trajectory(batch, gait, ticks, dt, seed, offset)
    out = []
    for t, s in enumerate(base):
        s = dict(s)
        s["q"] = np.ascontiguousarray(first["q"] + t * dt * first["qd"])
        s["qd"], s["a_des"] = first["qd"], first["a_des"]
        out.append(s)
    return out
