// free_gait_msgs/RobotState (ROS 1 wire format) -> the fields the controller's baseCommandCallback reads
// (balance_controller/src/ros_controller/ros_balance_controller.cpp:761-1083), one message per call.
// Message layout: free_gait_msgs/msg/{RobotState,LegMode,EndEffectorTarget}.msg plus the std ROS messages they
// embed (sensor_msgs/JointState, nav_msgs/Odometry, geometry_msgs/*Stamped, std_msgs/Header).
// ROS 1 serialisation: little-endian, no padding; string = uint32 length + bytes; T[] = uint32 count + items;
// T[N] = items only; time / duration = two 32-bit words; bool = one byte.
// Compiles for the device and, for the CPU-only tests, for the host.
#pragma once

#include <stdint.h>
#include <string.h>

#include "balance_core.hpp"

namespace qlamd {

enum WireStatus : int { kWireOk = 0, kWireTruncated = 1, kWireMissingField = 2 };
enum LegModeName : int { kModeOther = 0, kModeJoint = 1, kModeLegMode = 2, kModeCartesian = 3, kModeFootstep = 4 };

// Byte source: plain memory (host build, or a device pointer into global memory).  kOverread = false: the cursor
// never reads a byte at or beyond the message end.
struct PlainBytes {
  static constexpr bool kOverread = false;
  const uint8_t *p;
  QL_HD uint8_t u8(uint32_t at) const { return p[at]; }
  QL_HD uint32_t u32(uint32_t at) const { uint32_t v; memcpy(&v, p + at, 4); return v; }
  QL_HD double f64(uint32_t at) const { double v; memcpy(&v, p + at, 8); return v; }
};

// Positions are 32-bit and saturate at len + 1: once a length field points past the end the cursor stays "bad"
// and every later read returns 0 (or, for an over-readable source, harmless bytes of the staging window).
template <class Bytes>
struct WireCursor {
  Bytes p;
  uint32_t pos, len;

  QL_HD bool bad() const { return pos > len; }
  QL_HD void skip(uint32_t n) { pos = (n >= len + 1 - pos) ? len + 1 : pos + n; } // sticky: room is 0 once bad
  QL_HD void skip64(uint64_t n) { skip(n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n); }
  QL_HD uint32_t u32() {
    uint32_t v;
    if (Bytes::kOverread) v = p.u32(pos > len ? len : pos);
    else v = (pos <= len && len - pos >= 4) ? p.u32(pos) : 0u;
    skip(4);
    return bad() ? 0u : v;
  }
  QL_HD uint8_t u8() {
    uint8_t v;
    if (Bytes::kOverread) v = p.u8(pos > len ? len : pos);
    else v = (pos < len) ? p.u8(pos) : (uint8_t)0;
    skip(1);
    return bad() ? (uint8_t)0 : v;
  }
  QL_HD double f64() {
    double v;
    if (Bytes::kOverread) v = p.f64(pos > len ? len : pos);
    else v = (pos <= len && len - pos >= 8) ? p.f64(pos) : 0.0;
    skip(8);
    return bad() ? 0.0 : v;
  }
  QL_HD void skip_string() { const uint32_t n = u32(); skip(n); }
  QL_HD void skip_header() { skip(12); skip_string(); } // seq, stamp, frame_id
  // does the string of length slen that starts at `start` equal `lit` (length n)?  Only called when the string
  // lies inside the message.
  QL_HD bool string_is(const char *lit, uint32_t n, uint32_t slen, uint32_t start) const {
    if (slen != n) return false;
    bool same = true;
    QL_NOUNROLL for (uint32_t k = 0; k < n; k++) same = same && (p.u8(start + k) == (uint8_t)lit[k]);
    return same;
  }
};

struct RobotStateFields {
  double des_pos[3], des_quat[4], des_linvel[3], des_angvel[3]; // quaternion as (w, x, y, z)
  double joint_command[12];
  double foot_position[12], foot_velocity[12], foot_acceleration[12];
  double surface_normal[12], phase[4];
  uint8_t support_leg[4], leg_mode[4];
};

// sensor_msgs/JointState: keep position[0..2] (:802-812)
template <class Cur>
QL_HD void wire_joint_state(Cur &c, double *q, bool &missing) {
  c.skip_header();
  const uint32_t nn = c.u32();
  QL_NOUNROLL for (uint32_t k = 0; k < nn && !c.bad(); k++) c.skip_string();
  const uint32_t np = c.u32();
  if (np < 3) missing = true;
  const uint32_t keep = np < 3 ? np : 3;
  QL_NOUNROLL for (uint32_t k = 0; k < keep; k++) q[k] = c.f64();
  c.skip64(8 * ((uint64_t)np - keep));
  const uint32_t nv = c.u32(); c.skip64(8 * (uint64_t)nv);
  const uint32_t ne = c.u32(); c.skip64(8 * (uint64_t)ne);
}

// geometry_msgs/{Point,Vector3}Stamped[]: keep element 0 (:820-861) when `v0` is given, walk over the rest
template <class Cur>
QL_HD void wire_stamped_array(Cur &c, double *v0, bool &missing) {
  const uint32_t n = c.u32();
  if (n == 0 && v0) missing = true;
  QL_NOUNROLL for (uint32_t k = 0; k < n && !c.bad(); k++) {
    c.skip_header();
    if (k == 0 && v0) { v0[0] = c.f64(); v0[1] = c.f64(); v0[2] = c.f64(); }
    else c.skip(24);
  }
}

// `f` is indexed with run-time limb numbers on purpose (compact code: the walk is latency-bound, not issue-bound);
// on the device the caller keeps it in LDS, where run-time indexing is free.
template <class Bytes>
QL_HD int robot_state_unpack(const Bytes &msg, int64_t len, RobotStateFields &f) {
  if (len < 0 || len > 0x7FFFFFF0ll) return kWireTruncated;
  WireCursor<Bytes> c{msg, 0u, (uint32_t)len};
  bool missing = false;
  QL_NOUNROLL for (int l = 0; l < 4; l++) wire_joint_state(c, f.joint_command + 3 * l, missing); // lf, rf, rh, lh
  // nav_msgs/Odometry base_pose (:763-777)
  c.skip_header();
  c.skip_string();               // child_frame_id
  for (int k = 0; k < 3; k++) f.des_pos[k] = c.f64();
  {
    const double x = c.f64(), y = c.f64(), z = c.f64(), w = c.f64(); // geometry_msgs/Quaternion is x, y, z, w
    f.des_quat[0] = w; f.des_quat[1] = x; f.des_quat[2] = y; f.des_quat[3] = z;
  }
  c.skip(36 * 8);
  for (int k = 0; k < 3; k++) f.des_linvel[k] = c.f64();
  for (int k = 0; k < 3; k++) f.des_angvel[k] = c.f64();
  c.skip(36 * 8);
  // free_gait_msgs/LegMode x4 (:876-1078)
  QL_NOUNROLL for (int l = 0; l < 4; l++) {
    const uint32_t n = c.u32();
    const uint32_t start = c.pos;
    c.skip(n);
    int mode = kModeOther;
    if (!c.bad()) {
      if (c.string_is("joint", 5, n, start)) mode = kModeJoint;
      else if (c.string_is("leg_mode", 8, n, start)) mode = kModeLegMode;
      else if (c.string_is("cartesian", 9, n, start)) mode = kModeCartesian;
      else if (c.string_is("footstep", 8, n, start)) mode = kModeFootstep;
    }
    f.leg_mode[l] = (uint8_t)mode;
    f.support_leg[l] = c.u8() != 0;
    c.skip(8);                   // duration
    f.phase[l] = c.f64();
    c.skip_header();
    for (int k = 0; k < 3; k++) f.surface_normal[3 * l + k] = c.f64();
    c.skip(1);                   // ignore_for_pose_adaptation
  }
  // free_gait_msgs/EndEffectorTarget x4 (:816-861)
  QL_NOUNROLL for (int l = 0; l < 4; l++) {
    c.skip_string();
    wire_stamped_array(c, f.foot_position + 3 * l, missing);
    wire_stamped_array(c, f.foot_velocity + 3 * l, missing);
    wire_stamped_array(c, f.foot_acceleration + 3 * l, missing);
    wire_stamped_array(c, (double *)nullptr, missing); // target_force
    c.skip(8);                   // average_velocity
    c.skip_header();
    c.skip(24);                  // surface_normal
    c.skip(2);                   // ignore_contact, ignore_for_pose_adaptation
  }
  if (c.bad()) return kWireTruncated;
  return missing ? kWireMissingField : kWireOk;
}

} // namespace qlamd
