"""Writer for serialised free_gait_msgs/RobotState messages (ROS 1 wire format), the input of
qlamd_robot_state_unpack_batch / qlamd_full_tick_batch.

What StateRosPublisher puts on /desired_robot_state (free_gait_ros/src/StateRosPublisher.cpp:104-235,
free_gait_msgs/msg/RobotState.msg, LegMode.msg, EndEffectorTarget.msg): little-endian, unpadded, `string`
and `T[]` carry a uint32 count, time / duration are two 32-bit words, bool is one byte.  This module writes
the byte stream front to back with struct.pack (no schema table: tests/ros1_wire.py is the schema-driven
serialiser the tests check this one and the parsers against).
"""
import struct

import numpy as np

LEGS = ("lf", "rf", "rh", "lh")
MODE_NAMES = ("joint", "leg_mode", "cartesian", "footstep")  # ros_balance_controller.cpp:876-964
MODE_CODE = {name: k + 1 for k, name in enumerate(MODE_NAMES)}


class _Out:
    def __init__(self):
        self.parts = []

    def u32(self, v):
        self.parts.append(struct.pack("<I", int(v) & 0xFFFFFFFF))

    def i32(self, v):
        self.parts.append(struct.pack("<i", int(v)))

    def u8(self, v):
        self.parts.append(struct.pack("<B", int(v)))

    def f64(self, *vs):
        self.parts.append(struct.pack("<%dd" % len(vs), *[float(v) for v in vs]))

    def string(self, s):
        raw = s.encode()
        self.u32(len(raw))
        self.parts.append(raw)

    def header(self, h):
        # std_msgs/Header: seq, stamp (secs, nsecs), frame_id
        self.u32(h.get("seq", 0))
        self.u32(h.get("secs", 0))
        self.u32(h.get("nsecs", 0))
        self.string(h.get("frame_id", ""))

    def stamped3(self, h, v):
        # geometry_msgs/PointStamped and Vector3Stamped share the layout
        self.header(h)
        self.f64(v[0], v[1], v[2])

    def bytes(self):
        return b"".join(self.parts)


def pack_robot_state(f, layout=None):
    """Serialise one RobotState.

    f: dict with des_pos[3], des_quat[4] (w,x,y,z), des_linvel[3], des_angvel[3], joint_command[12],
       foot_position[12], foot_velocity[12], foot_acceleration[12], surface_normal[12], phase[4],
       support_leg[4], mode_name[4] (strings; an unknown name leaves the leg's mode as it is).
    layout: optional dict of everything that moves field offsets without carrying payload --
       frame_id (str), joint_names[4] (lists of str, >= 3), extra_positions[4], velocities[4], efforts[4] (lists of
       float), extra_targets[4] (dict kind -> list of 3-vectors), target_name[4], child_frame_id.
    """
    L = layout or {}
    hdr = {"frame_id": L.get("frame_id", "")}
    o = _Out()
    jc = np.asarray(f["joint_command"], dtype=np.float64).ravel()
    for l in range(4):  # sensor_msgs/JointState x 4
        names = L.get("joint_names", [["", "", ""]] * 4)[l]
        extra = L.get("extra_positions", [[]] * 4)[l]
        vel = L.get("velocities", [[]] * 4)[l]
        eff = L.get("efforts", [[]] * 4)[l]
        o.header(hdr)
        o.u32(len(names))
        for n in names:
            o.string(n)
        o.u32(3 + len(extra))
        o.f64(*jc[3 * l:3 * l + 3], *extra)
        o.u32(len(vel))
        if len(vel):
            o.f64(*vel)
        o.u32(len(eff))
        if len(eff):
            o.f64(*eff)
    # nav_msgs/Odometry
    o.header(hdr)
    o.string(L.get("child_frame_id", ""))
    q = f["des_quat"]
    o.f64(*f["des_pos"])
    o.f64(q[1], q[2], q[3], q[0])  # geometry_msgs/Quaternion is x, y, z, w
    o.f64(*L.get("pose_covariance", [0.0] * 36))
    o.f64(*f["des_linvel"])
    o.f64(*f["des_angvel"])
    o.f64(*L.get("twist_covariance", [0.0] * 36))
    sn = np.asarray(f["surface_normal"], dtype=np.float64).ravel()
    for l in range(4):  # free_gait_msgs/LegMode x 4
        o.string(f["mode_name"][l])
        o.u8(f["support_leg"][l])
        o.i32(0)
        o.i32(0)  # duration
        o.f64(f["phase"][l])
        o.stamped3(hdr, sn[3 * l:3 * l + 3])
        o.u8(0)
    fp = np.asarray(f["foot_position"], dtype=np.float64).ravel()
    fv = np.asarray(f["foot_velocity"], dtype=np.float64).ravel()
    fa = np.asarray(f["foot_acceleration"], dtype=np.float64).ravel()
    for l in range(4):  # free_gait_msgs/EndEffectorTarget x 4
        more = L.get("extra_targets", [{}] * 4)[l]
        o.string(L.get("target_name", [""] * 4)[l])
        for kind, first in (("position", fp), ("velocity", fv), ("acceleration", fa)):
            rest = more.get(kind, [])
            o.u32(1 + len(rest))
            o.stamped3(hdr, first[3 * l:3 * l + 3])
            for v in rest:
                o.stamped3(hdr, v)
        force = more.get("force", [])
        o.u32(len(force))
        for v in force:
            o.stamped3(hdr, v)
        o.f64(0.0)  # average_velocity
        o.stamped3(hdr, (0.0, 0.0, 1.0))
        o.u8(0)
        o.u8(0)
    return o.bytes()


def random_layout(rng):
    """A layout with its own string lengths and array counts (every message gets different field offsets)."""
    word = lambda: "".join(rng.choice(list("abcdefgh_/0123"), int(rng.integers(0, 12))))  # noqa: E731
    vecs = lambda: [rng.normal(size=3) for _ in range(int(rng.integers(0, 3)))]  # noqa: E731
    return dict(frame_id=word(), child_frame_id=word(),
                joint_names=[[word() for _ in range(3 + int(rng.integers(0, 3)))] for _ in range(4)],
                extra_positions=[list(rng.normal(size=int(rng.integers(0, 3)))) for _ in range(4)],
                velocities=[list(rng.normal(size=int(rng.integers(0, 4)))) for _ in range(4)],
                efforts=[list(rng.normal(size=int(rng.integers(0, 4)))) for _ in range(4)],
                target_name=[word() for _ in range(4)],
                extra_targets=[dict(position=vecs(), velocity=vecs(), acceleration=vecs(), force=vecs()) for _ in range(4)])


def pack_batch(messages):
    """Concatenate serialised messages: (uint8 blob, int64 offsets[B + 1]) as the C-ABI takes them."""
    off = np.zeros(len(messages) + 1, np.int64)
    off[1:] = np.cumsum([len(m) for m in messages])
    return np.frombuffer(b"".join(messages), np.uint8).copy(), off
