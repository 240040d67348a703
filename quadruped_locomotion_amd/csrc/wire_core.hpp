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

struct WireCursor {
  const uint8_t *p;
  int64_t pos, len;
  bool bad;

  QL_HD bool need(int64_t n) {
    if (bad || n < 0 || pos + n > len) { bad = true; return false; }
    return true;
  }
  QL_HD uint32_t u32() {
    if (!need(4)) return 0;
    uint32_t v;
    memcpy(&v, p + pos, 4);
    pos += 4;
    return v;
  }
  QL_HD uint8_t u8() {
    if (!need(1)) return 0;
    return p[pos++];
  }
  QL_HD double f64() {
    if (!need(8)) return 0.0;
    double v;
    memcpy(&v, p + pos, 8);
    pos += 8;
    return v;
  }
  QL_HD void skip(int64_t n) { if (need(n)) pos += n; }
  QL_HD void skip_string() { const uint32_t n = u32(); skip(n); }
  QL_HD void skip_header() { skip(12); skip_string(); } // seq, stamp, frame_id
  QL_HD void vec3(double v[3]) { v[0] = f64(); v[1] = f64(); v[2] = f64(); }
  // does the string at the cursor equal `lit` (length n)?  Consumes the string.
  QL_HD bool string_is(const char *lit, uint32_t n, uint32_t slen, int64_t start) const {
    if (slen != n) return false;
    for (uint32_t k = 0; k < n; k++)
      if (p[start + k] != (uint8_t)lit[k]) return false;
    return true;
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
QL_HD void wire_joint_state(WireCursor &c, double q[3], bool &missing) {
  c.skip_header();
  const uint32_t nn = c.u32();
  for (uint32_t k = 0; k < nn && !c.bad; k++) c.skip_string();
  const uint32_t np = c.u32();
  if (np < 3) missing = true;
  for (uint32_t k = 0; k < np && !c.bad; k++) {
    const double v = c.f64();
    if (k < 3) q[k] = v;
  }
  const uint32_t nv = c.u32(); c.skip(8 * (int64_t)nv);
  const uint32_t ne = c.u32(); c.skip(8 * (int64_t)ne);
}

// geometry_msgs/{Point,Vector3}Stamped[]: keep element 0 (:820-861), skip the rest
QL_HD void wire_stamped_array(WireCursor &c, double v0[3], bool required, bool &missing) {
  const uint32_t n = c.u32();
  if (n == 0 && required) missing = true;
  for (uint32_t k = 0; k < n && !c.bad; k++) {
    c.skip_header();
    double v[3];
    c.vec3(v);
    if (k == 0) { v0[0] = v[0]; v0[1] = v[1]; v0[2] = v[2]; }
  }
}

QL_HD int robot_state_unpack(const uint8_t *msg, int64_t len, RobotStateFields &f) {
  WireCursor c{msg, 0, len, false};
  bool missing = false;
  for (int l = 0; l < 4; l++) wire_joint_state(c, f.joint_command + 3 * l, missing); // lf, rf, rh, lh
  // nav_msgs/Odometry base_pose (:763-777)
  c.skip_header();
  c.skip_string();               // child_frame_id
  c.vec3(f.des_pos);
  {
    const double x = c.f64(), y = c.f64(), z = c.f64(), w = c.f64(); // geometry_msgs/Quaternion is x, y, z, w
    f.des_quat[0] = w; f.des_quat[1] = x; f.des_quat[2] = y; f.des_quat[3] = z;
  }
  c.skip(36 * 8);
  c.vec3(f.des_linvel);
  c.vec3(f.des_angvel);
  c.skip(36 * 8);
  // free_gait_msgs/LegMode x4 (:876-1078)
  for (int l = 0; l < 4; l++) {
    const uint32_t n = c.u32();
    const int64_t start = c.pos;
    c.skip(n);
    int mode = kModeOther;
    if (!c.bad) {
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
    c.vec3(f.surface_normal + 3 * l);
    c.skip(1);                   // ignore_for_pose_adaptation
  }
  // free_gait_msgs/EndEffectorTarget x4 (:816-861)
  for (int l = 0; l < 4; l++) {
    c.skip_string();
    double force[3];
    wire_stamped_array(c, f.foot_position + 3 * l, true, missing);
    wire_stamped_array(c, f.foot_velocity + 3 * l, true, missing);
    wire_stamped_array(c, f.foot_acceleration + 3 * l, true, missing);
    wire_stamped_array(c, force, false, missing);
    c.skip(8);                   // average_velocity
    c.skip_header();
    c.skip(24);                  // surface_normal
    c.skip(2);                   // ignore_contact, ignore_for_pose_adaptation
  }
  if (c.bad) return kWireTruncated;
  return missing ? kWireMissingField : kWireOk;
}

} // namespace qlamd
