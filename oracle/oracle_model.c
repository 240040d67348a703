/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_model.h). */
#include "oracle_model.h"

#include <math.h>
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
