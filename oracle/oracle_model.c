/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_model.h). */
#include "oracle_model.h"

#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <string.h>

#include "qlamd_robot_constants.h"

typedef struct { double R[9]; double p[3]; } frame_t;

static void mat3_mul(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double acc = 0.0;
      for (int k = 0; k < 3; k++) acc += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = acc;
    }
}

static void mat3_vec(const double *A, const double *v, double *out) {
  for (int i = 0; i < 3; i++) out[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}

static void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* URDF fixed-axis rpy: R = Rz(yaw) Ry(pitch) Rx(roll)   */
static void rpy_to_mat(const double rpy[3], double *R) {
  double cr = cos(rpy[0]), sr = sin(rpy[0]);
  double cp = cos(rpy[1]), sp = sin(rpy[1]);
  double cy = cos(rpy[2]), sy = sin(rpy[2]);
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

/* KDL Segment::pose(q) with kdl_parser's joint placement:
 *   T_seg(q) = Trans(xyz) * R0(rpy) * Rz(q)   (revolute about local z),
 *   T_seg    = Trans(xyz) * R0(rpy)           (fixed). */
static void segment_pose(int leg, int seg, double q, frame_t *T) {
  double R0[9];
  rpy_to_mat(QLAMD_JOINT_RPY[leg][seg], R0);
  if (QLAMD_SEG_REVOLUTE[leg][seg]) {
    double c = cos(q), s = sin(q);
    double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
    mat3_mul(R0, Rz, T->R);
  } else {
    memcpy(T->R, R0, sizeof(R0));
  }
  memcpy(T->p, QLAMD_JOINT_XYZ[leg][seg], 3 * sizeof(double));
}

/* ChainFkSolverPos_recursive: left-to-right product of the segment poses.
 * cum[k] = pose of segment k's tip (= link k frame) in the base frame. */
static void chain_frames(int leg, const double q[3], frame_t cum[4]) {
  frame_t cur;
  memset(&cur, 0, sizeof(cur));
  cur.R[0] = cur.R[4] = cur.R[8] = 1.0;
  for (int k = 0; k < 4; k++) {
    frame_t seg, nxt;
    segment_pose(leg, k, k < 3 ? q[k] : 0.0, &seg);
    mat3_mul(cur.R, seg.R, nxt.R);
    double rp[3];
    mat3_vec(cur.R, seg.p, rp);
    for (int i = 0; i < 3; i++) nxt.p[i] = cur.p[i] + rp[i];
    cum[k] = nxt;
    cur = nxt;
  }
}

void oracle_leg_fk(int leg, const double q[3], double p[3], double R[9]) {
  frame_t cum[4];
  chain_frames(leg, q, cum);
  memcpy(p, cum[3].p, 3 * sizeof(double));
  if (R) memcpy(R, cum[3].R, 9 * sizeof(double));
}

/* ChainJntToJacSolver::JntToJac: column i = unit twist of joint i expressed in
 * the base frame with the reference point moved to the chain tip:
 *   v_i = z_i x (p_tip - p_i),  z_i = axis of joint i in base = 3rd column of
 *   the link-i frame (Rz(q) leaves z invariant), p_i = origin of link-i frame. */
void oracle_leg_jacobian(int leg, const double q[3], double J[9]) {
  frame_t cum[4];
  chain_frames(leg, q, cum);
  for (int i = 0; i < 3; i++) {
    double z[3] = {cum[i].R[2], cum[i].R[5], cum[i].R[8]};
    double d[3] = {cum[3].p[0] - cum[i].p[0], cum[3].p[1] - cum[i].p[1], cum[3].p[2] - cum[i].p[2]};
    double v[3];
    cross3(z, d, v);
    for (int r = 0; r < 3; r++) J[r * 3 + i] = v[r];
  }
}

static void link_coms(int leg, const frame_t cum[4], double c[4][3]) {
  for (int k = 0; k < 4; k++) {
    double rc[3];
    mat3_vec(cum[k].R, QLAMD_LINK_COM[leg][k], rc);
    for (int i = 0; i < 3; i++) c[k][i] = cum[k].p[i] + rc[i];
  }
}

/* ChainDynParam::JntToGravity = ChainIdSolver_RNE with zero joint rates and
 * base acceleration -g.  With no velocities every link's inertial force is
 * m_k * (-g) at its centre of mass; the torque at joint i is the projection
 * on z_i of the moments of all distal links (segments i..3; the fixed foot
 * link carries mass too). */
void oracle_leg_gravity(int leg, const double q[3], const double g[3], double G[3]) {
  frame_t cum[4];
  double c[4][3];
  chain_frames(leg, q, cum);
  link_coms(leg, cum, c);
  for (int i = 0; i < 3; i++) {
    double z[3] = {cum[i].R[2], cum[i].R[5], cum[i].R[8]};
    double acc = 0.0;
    for (int k = i; k < 4; k++) {
      double d[3] = {c[k][0] - cum[i].p[0], c[k][1] - cum[i].p[1], c[k][2] - cum[i].p[2]};
      double v[3];
      cross3(z, d, v);
      acc += QLAMD_LINK_MASS[leg][k] * (g[0] * v[0] + g[1] * v[1] + g[2] * v[2]);
    }
    G[i] = -acc;
  }
}

double oracle_leg_potential(int leg, const double q[3], const double g[3]) {
  frame_t cum[4];
  double c[4][3];
  chain_frames(leg, q, cum);
  link_coms(leg, cum, c);
  double U = 0.0;
  for (int k = 0; k < 4; k++)
    U -= QLAMD_LINK_MASS[leg][k] * (g[0] * c[k][0] + g[1] * c[k][1] + g[2] * c[k][2]);
  return U;
}

/* QuadrupedKinematics::MapToPI (quadrupedkinematics.cpp:554-563): note 2 pi - q, not q - 2 pi, above pi */
static double map_to_pi(double q) {
  double out = q;
  if (q > M_PI) out = 2 * M_PI - q;
  if (q < -M_PI) out = 2 * M_PI + q;
  return out;
}

/* QuadrupedKinematics::InverseKinematicsSolve (quadrupedkinematics.cpp:377-483).
 * hip frame = first segment's frame at q = 0 (setHipPoseInBase, :109-122; KDL Segment::getFrameToTip() =
 * joint.pose(0) * f_tip = the URDF joint origin, xyz + rpy).  config = row of `results` the reference picks:
 * 0 "OUT_LEFT", 1 "IN_RIGHT", 2 "IN_LEFT", 3 "OUT_RIGHT" (:466-473).  geom = {d, l1, l2}; the reference
 * hard-codes {0.1, 0.25, 0.25} (:383-385).  A left-arm row with a == 0 is never written by the reference
 * (:420-429); it is NaN here.  Returns 1 when all three angles are finite (:478-483). */
int oracle_leg_ik(int leg, const double p_base[3], int config, const double geom[3], double q_out[3]) {
  const double d = geom[0], l1 = geom[1], l2 = geom[2];
  double R0[9], rel[3], ph[3];
  rpy_to_mat(QLAMD_JOINT_RPY[leg][0], R0);
  for (int i = 0; i < 3; i++) rel[i] = p_base[i] - QLAMD_JOINT_XYZ[leg][0][i];
  for (int i = 0; i < 3; i++) ph[i] = R0[i] * rel[0] + R0[3 + i] * rel[1] + R0[6 + i] * rel[2]; /* R0^T rel */
  const double px = ph[0], py = ph[1], pz = ph[2];
  double cos_theta3 = (l2 * l2 + l1 * l1 - ((px * px + py * py + pz * pz) - d * d)) / 2 / l1 / l2;
  if (cos_theta3 < -1) cos_theta3 = -1;
  if (cos_theta3 > 1) cos_theta3 = 1;
  const double theta3 = (config < 2) ? M_PI - acos(cos_theta3) : -M_PI + acos(cos_theta3);
  const double alpha = atan2(py, px);
  const double rxy = sqrt(fabs(px * px + py * py - d * d));
  const double beta1 = atan2(d, rxy), beta2 = atan2(-d, -rxy);
  const double q3 = map_to_pi(theta3);
  const double b = atan2(l2 * sin(q3), l1 + l2 * cos(q3));
  double q1, q2;
  if ((config & 1) == 0) { /* left arm, rows 0 and 2 */
    q1 = map_to_pi(alpha - beta1);
    const double a = atan2(pz, -rxy);
    if (a > 0) q2 = map_to_pi(a - b - M_PI);
    else if (a < 0) q2 = map_to_pi(a - b + M_PI);
    else q2 = (double)NAN;
  } else {                 /* right arm, rows 1 and 3 */
    q1 = map_to_pi(alpha + beta2);
    const double a = atan2(pz, rxy);
    q2 = map_to_pi(a - b + M_PI);
  }
  q_out[0] = q1; q_out[1] = q2; q_out[2] = q3;
  return !isnan(q1) && !isnan(q2) && !isnan(q3);
}
