/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_wire.h). */
#include "oracle_wire.h"

#include <string.h>

typedef struct { const uint8_t *b; size_t n, at; int fail; } rd_t;

static int have(rd_t *r, size_t k) {
  if (r->fail || r->n - r->at < k || r->at > r->n) { r->fail = 1; return 0; }
  return 1;
}
static uint32_t rd_u32(rd_t *r) {
  if (!have(r, 4)) return 0;
  const uint8_t *p = r->b + r->at;
  r->at += 4;
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static double rd_f64(rd_t *r) {
  if (!have(r, 8)) return 0.0;
  uint64_t u = 0;
  for (int k = 7; k >= 0; k--) u = (u << 8) | r->b[r->at + (size_t)k];
  r->at += 8;
  double d;
  memcpy(&d, &u, 8);
  return d;
}
static void rd_skip(rd_t *r, size_t k) { if (have(r, k)) r->at += k; }
static void rd_header(rd_t *r) { rd_skip(r, 12); rd_skip(r, rd_u32(r)); }
static void rd_xyz(rd_t *r, double *v) { for (int i = 0; i < 3; i++) v[i] = rd_f64(r); }

static void rd_first_stamped(rd_t *r, double *first, int required, int *missing) {
  const uint32_t count = rd_u32(r);
  if (count == 0 && required) *missing = 1;
  for (uint32_t k = 0; k < count && !r->fail; k++) {
    double v[3];
    rd_header(r);
    rd_xyz(r, v);
    if (k == 0) memcpy(first, v, sizeof(v));
  }
}

int oracle_robot_state_unpack(const uint8_t *msg, size_t len, oracle_robot_state_fields *o) {
  rd_t r = {msg, len, 0, 0};
  int missing = 0;
  /* sensor_msgs/JointState {lf,rf,rh,lh}_leg_joints */
  for (int l = 0; l < 4; l++) {
    rd_header(&r);
    for (uint32_t k = rd_u32(&r); k > 0 && !r.fail; k--) rd_skip(&r, rd_u32(&r)); /* name[] */
    const uint32_t np = rd_u32(&r);
    if (np < 3) missing = 1;
    for (uint32_t k = 0; k < np && !r.fail; k++) {
      const double v = rd_f64(&r);
      if (k < 3) o->joint_command[3 * l + (int)k] = v;
    }
    rd_skip(&r, 8 * (size_t)rd_u32(&r)); /* velocity[] */
    rd_skip(&r, 8 * (size_t)rd_u32(&r)); /* effort[] */
  }
  /* nav_msgs/Odometry base_pose */
  rd_header(&r);
  rd_skip(&r, rd_u32(&r)); /* child_frame_id */
  rd_xyz(&r, o->des_pos);
  double q[4];
  for (int i = 0; i < 4; i++) q[i] = rd_f64(&r); /* x y z w */
  o->des_quat[0] = q[3]; o->des_quat[1] = q[0]; o->des_quat[2] = q[1]; o->des_quat[3] = q[2];
  rd_skip(&r, 288);
  rd_xyz(&r, o->des_linvel);
  rd_xyz(&r, o->des_angvel);
  rd_skip(&r, 288);
  /* free_gait_msgs/LegMode */
  static const char *names[5] = {"", "joint", "leg_mode", "cartesian", "footstep"};
  for (int l = 0; l < 4; l++) {
    const uint32_t n = rd_u32(&r);
    const size_t at = r.at;
    rd_skip(&r, n);
    o->leg_mode[l] = 0;
    for (int m = 1; m < 5 && !r.fail; m++)
      if (strlen(names[m]) == n && memcmp(msg + at, names[m], n) == 0) o->leg_mode[l] = (uint8_t)m;
    if (have(&r, 1)) o->support_leg[l] = msg[r.at++] != 0;
    rd_skip(&r, 8); /* duration */
    o->phase[l] = rd_f64(&r);
    rd_header(&r);
    rd_xyz(&r, o->surface_normal + 3 * l);
    rd_skip(&r, 1);
  }
  /* free_gait_msgs/EndEffectorTarget */
  for (int l = 0; l < 4; l++) {
    double unused[3];
    rd_skip(&r, rd_u32(&r)); /* name */
    rd_first_stamped(&r, o->foot_position + 3 * l, 1, &missing);
    rd_first_stamped(&r, o->foot_velocity + 3 * l, 1, &missing);
    rd_first_stamped(&r, o->foot_acceleration + 3 * l, 1, &missing);
    rd_first_stamped(&r, unused, 0, &missing);
    rd_skip(&r, 8);
    rd_header(&r);
    rd_skip(&r, 24 + 2);
  }
  if (r.fail) return 1;
  return missing ? 2 : 0;
}
