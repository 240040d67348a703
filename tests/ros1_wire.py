"""A small ROS 1 message serialiser written from the message definitions (test helper): the independent side of
the wire-format tests.  Schema = the .msg files of free_gait_msgs (RobotState, LegMode, EndEffectorTarget) and
the standard messages they embed; rules = roscpp serialisation (little-endian, unpadded; string and T[] carry a
uint32 length; T[N] does not; time/duration are two 32-bit words; bool is one byte)."""
import struct

SCHEMA = {
    "std_msgs/Header": [("uint32", "seq"), ("time", "stamp"), ("string", "frame_id")],
    "geometry_msgs/Point": [("float64", "x"), ("float64", "y"), ("float64", "z")],
    "geometry_msgs/Vector3": [("float64", "x"), ("float64", "y"), ("float64", "z")],
    "geometry_msgs/Quaternion": [("float64", "x"), ("float64", "y"), ("float64", "z"), ("float64", "w")],
    "geometry_msgs/Pose": [("geometry_msgs/Point", "position"), ("geometry_msgs/Quaternion", "orientation")],
    "geometry_msgs/Twist": [("geometry_msgs/Vector3", "linear"), ("geometry_msgs/Vector3", "angular")],
    "geometry_msgs/PoseWithCovariance": [("geometry_msgs/Pose", "pose"), ("float64[36]", "covariance")],
    "geometry_msgs/TwistWithCovariance": [("geometry_msgs/Twist", "twist"), ("float64[36]", "covariance")],
    "geometry_msgs/PointStamped": [("std_msgs/Header", "header"), ("geometry_msgs/Point", "point")],
    "geometry_msgs/Vector3Stamped": [("std_msgs/Header", "header"), ("geometry_msgs/Vector3", "vector")],
    "nav_msgs/Odometry": [("std_msgs/Header", "header"), ("string", "child_frame_id"),
                          ("geometry_msgs/PoseWithCovariance", "pose"), ("geometry_msgs/TwistWithCovariance", "twist")],
    "sensor_msgs/JointState": [("std_msgs/Header", "header"), ("string[]", "name"), ("float64[]", "position"),
                               ("float64[]", "velocity"), ("float64[]", "effort")],
    "free_gait_msgs/LegMode": [("string", "name"), ("bool", "support_leg"), ("duration", "duration"), ("float64", "phase"),
                               ("geometry_msgs/Vector3Stamped", "surface_normal"), ("bool", "ignore_for_pose_adaptation")],
    "free_gait_msgs/EndEffectorTarget": [
        ("string", "name"), ("geometry_msgs/PointStamped[]", "target_position"),
        ("geometry_msgs/Vector3Stamped[]", "target_velocity"), ("geometry_msgs/Vector3Stamped[]", "target_acceleration"),
        ("geometry_msgs/Vector3Stamped[]", "target_force"), ("float64", "average_velocity"),
        ("geometry_msgs/Vector3Stamped", "surface_normal"), ("bool", "ignore_contact"), ("bool", "ignore_for_pose_adaptation")],
    "free_gait_msgs/RobotState": (
        [("sensor_msgs/JointState", f"{l}_leg_joints") for l in ("lf", "rf", "rh", "lh")] + [("nav_msgs/Odometry", "base_pose")] +
        [("free_gait_msgs/LegMode", f"{l}_leg_mode") for l in ("lf", "rf", "rh", "lh")] +
        [("free_gait_msgs/EndEffectorTarget", f"{l}_target") for l in ("lf", "rf", "rh", "lh")]),
}
PRIMITIVE = {"uint32": "<I", "int32": "<i", "float64": "<d", "bool": "<B", "uint8": "<B"}


def serialize(typ, value):
    if typ.endswith("]"):
        base, dim = typ[:-1].split("[")
        items = list(value) if value is not None else []
        if dim:
            n = int(dim)
            items = (items + [None] * n)[:n]
            head = b""
        else:
            head = struct.pack("<I", len(items))
        return head + b"".join(serialize(base, v) for v in items)
    if typ in PRIMITIVE:
        return struct.pack(PRIMITIVE[typ], value if value is not None else 0)
    if typ == "string":
        raw = (value or "").encode()
        return struct.pack("<I", len(raw)) + raw
    if typ in ("time", "duration"):
        secs, nsecs = value if value is not None else (0, 0)
        return struct.pack("<ii" if typ == "duration" else "<II", secs, nsecs)
    value = value or {}
    return b"".join(serialize(t, value.get(name)) for t, name in SCHEMA[typ])


def xyz(v):
    return dict(x=float(v[0]), y=float(v[1]), z=float(v[2]))


def stamped(kind, v, frame="odom", seq=0):
    return {"header": dict(seq=seq, stamp=(12, 34), frame_id=frame), kind: xyz(v)}
