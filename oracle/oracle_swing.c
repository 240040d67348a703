/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_swing.h). */
#include "oracle_swing.h"

#include <math.h>
#include <string.h>

#include "oracle_model.h"
#include "qlamd_robot_constants.h"

static void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static void mat3_mul(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void mat3_vec(const double *A, const double *v, double *o) {
  for (int i = 0; i < 3; i++) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
static void rpy_to_mat(const double rpy[3], double *R) {
  double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

/* Recursive Newton-Euler in base coordinates for a fixed-base chain; the fixed foot link rides on
 * link 3 (RBDL merges fixed bodies into their movable parent).  The chain is given segment by segment as the URDF
 * gives it (joint origin xyz / rpy, link mass, centre of mass and inertia about it in the link frame). */
void oracle_chain_rnea(const double joint_xyz[4][3], const double joint_rpy[4][3], const double link_mass[4],
                       const double link_com[4][3], const double link_inertia[4][6], const double q[3],
                       const double qd[3], const double qdd[3], const double g[3], double tau[3]) {
  double R[4][9], p[4][3];
  {
    double Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pc[3] = {0, 0, 0};
    for (int k = 0; k < 4; k++) {
      double R0[9], Rs[9], Rn[9], rp[3];
      rpy_to_mat(joint_rpy[k], R0);
      if (k < 3) {
        const double c = cos(q[k]), s = sin(q[k]);
        const double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
        mat3_mul(R0, Rz, Rs);
      } else {
        memcpy(Rs, R0, sizeof(Rs));
      }
      mat3_vec(Rc, joint_xyz[k], rp);
      for (int i = 0; i < 3; i++) pc[i] += rp[i];
      mat3_mul(Rc, Rs, Rn);
      memcpy(Rc, Rn, sizeof(Rc));
      memcpy(R[k], Rn, sizeof(Rn));
      memcpy(p[k], pc, sizeof(pc));
    }
  }
  /* forward pass: angular velocity / acceleration of links 0..2, linear acceleration of the joint origins */
  double w[3][3], al[3][3], a[3][3], z[3][3];
  for (int i = 0; i < 3; i++) {
    for (int k = 0; k < 3; k++) z[i][k] = R[i][3 * k + 2];
    double wprev[3] = {0, 0, 0}, aprev[3] = {0, 0, 0}, accprev[3] = {-g[0], -g[1], -g[2]}, d[3] = {0, 0, 0};
    if (i > 0) {
      memcpy(wprev, w[i - 1], sizeof(wprev));
      memcpy(aprev, al[i - 1], sizeof(aprev));
      memcpy(accprev, a[i - 1], sizeof(accprev));
      for (int k = 0; k < 3; k++) d[k] = p[i][k] - p[i - 1][k];
    }
    double zq[3] = {qd[i] * z[i][0], qd[i] * z[i][1], qd[i] * z[i][2]}, wxzq[3], axd[3], wxd[3], wxwxd[3];
    cross3(wprev, zq, wxzq);
    cross3(aprev, d, axd);
    cross3(wprev, d, wxd);
    cross3(wprev, wxd, wxwxd);
    for (int k = 0; k < 3; k++) {
      w[i][k] = wprev[k] + zq[k];
      al[i][k] = aprev[k] + qdd[i] * z[i][k] + wxzq[k];
      a[i][k] = accprev[k] + axd[k] + wxwxd[k];
    }
  }
  /* bodies: links 0,1,2 and the foot (rigid on link 2) */
  double F[4][3], N[4][3], c[4][3];
  for (int b = 0; b < 4; b++) {
    const int i = b < 3 ? b : 2;
    double rc[3], dc[3], axd[3], wxd[3], wxwxd[3], acc[3];
    mat3_vec(R[b], link_com[b], rc);
    for (int k = 0; k < 3; k++) { c[b][k] = p[b][k] + rc[k]; dc[k] = c[b][k] - p[i][k]; }
    cross3(al[i], dc, axd);
    cross3(w[i], dc, wxd);
    cross3(w[i], wxd, wxwxd);
    for (int k = 0; k < 3; k++) { acc[k] = a[i][k] + axd[k] + wxwxd[k]; F[b][k] = link_mass[b] * acc[k]; }
    const double *I6 = link_inertia[b]; /* ixx ixy ixz iyy iyz izz */
    const double Il[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
    double T[9], Rt[9], Ib[9], Ia[3], Iw[3], wIw[3];
    for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++) Rt[3 * r + s] = R[b][3 * s + r];
    mat3_mul(R[b], Il, T);
    mat3_mul(T, Rt, Ib);
    mat3_vec(Ib, al[i], Ia);
    mat3_vec(Ib, w[i], Iw);
    cross3(w[i], Iw, wIw);
    for (int k = 0; k < 3; k++) N[b][k] = Ia[k] + wIw[k];
  }
  for (int i = 0; i < 3; i++) {
    double acc = 0.0;
    for (int b = i; b < 4; b++) {
      double d[3] = {c[b][0] - p[i][0], c[b][1] - p[i][1], c[b][2] - p[i][2]}, m[3];
      cross3(d, F[b], m);
      for (int k = 0; k < 3; k++) acc += z[i][k] * (N[b][k] + m[k]);
    }
    tau[i] = acc;
  }
}

void oracle_chain_rnea_rows(long rows, const double joint_xyz[4][3], const double joint_rpy[4][3],
                            const double link_mass[4], const double link_com[4][3], const double link_inertia[4][6],
                            const double *q, const double *qd, const double *qdd, const double g[3], double *tau) {
  for (long i = 0; i < rows; i++)
    oracle_chain_rnea(joint_xyz, joint_rpy, link_mass, link_com, link_inertia, q + 3 * i, qd + 3 * i, qdd + 3 * i, g,
                      tau + 3 * i);
}

void oracle_leg_rnea(int leg, const double q[3], const double qd[3], const double qdd[3], const double g[3],
                     double tau[3]) {
  oracle_chain_rnea(QLAMD_JOINT_XYZ[leg], QLAMD_JOINT_RPY[leg], QLAMD_LINK_MASS[leg], QLAMD_LINK_COM[leg],
                    QLAMD_LINK_INERTIA[leg], q, qd, qdd, g, tau);
}

void oracle_swing_default_params(oracle_swing_params *p) {
  for (int i = 0; i < 3; i++) { p->kp[i] = 300.0; p->kd[i] = 20.0; } /* controller_gains.yaml:42-51 */
  p->period = 0.0025;
  p->accel_window = 10.0;
  p->accel_scale = 0.5;
  p->gravity = 9.81;
}

void oracle_swing_leg_torque(const oracle_swing_params *P, int leg, const double q_id[3], const double q[3],
                             const double qd[3], const double qd_oldest[3], const double target_pos[3],
                             const double target_vel[3], double tau[3]) {
  double qdd[3], g[3] = {0.0, 0.0, -P->gravity}, tid[3], J[9], foot[3];
  for (int i = 0; i < 3; i++) qdd[i] = P->accel_scale * ((qd[i] - qd_oldest[i]) / (P->period * P->accel_window));
  oracle_leg_rnea(leg, q_id, qd, qdd, g, tid);
  oracle_leg_jacobian(leg, q, J);
  oracle_leg_fk(leg, q, foot, NULL);
  double f[3];
  for (int r = 0; r < 3; r++) {
    const double v = J[3 * r] * qd[0] + J[3 * r + 1] * qd[1] + J[3 * r + 2] * qd[2]; /* quadruped_state.cpp:315-319 */
    f[r] = P->kp[r] * (target_pos[r] - foot[r]) + P->kd[r] * (target_vel[r] - v);
  }
  for (int j = 0; j < 3; j++) tau[j] = (J[j] * f[0] + J[3 + j] * f[1] + J[6 + j] * f[2]) + tid[j];
}

void oracle_pid_default_params(oracle_pid_params *p) {
  for (int j = 0; j < 12; j++) { /* balance_controller/config/control.yaml:18-29 */
    p->p[j] = 300.0; p->i[j] = 0.01; p->d[j] = 3.0;
    p->i_max[j] = 0.0; p->i_min[j] = 0.0;      /* no i_clamp parameter: control_toolbox defaults */
    p->lower[j] = -3.0; p->upper[j] = 3.0;      /* quadruped_model.urdf:53-57 and siblings */
  }
  p->antiwindup = 0;
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

static double pid_command(const oracle_pid_params *g, int j, double error, double dt, double *e_last, double *e_int) {
  if (dt == 0.0 || isnan(error) || isinf(error)) return 0.0;
  double error_dot = 0.0;
  if (dt > 0.0) {
    error_dot = (error - *e_last) / dt;
    *e_last = error;
  }
  if (isnan(error_dot) || isinf(error_dot)) return 0.0;
  const double p_term = g->p[j] * error;
  *e_int += dt * error;
  if (g->antiwindup && g->i[j] != 0.0) {
    const double a = g->i_min[j] / g->i[j], b = g->i_max[j] / g->i[j];
    *e_int = clampd(*e_int, a < b ? a : b, a < b ? b : a);
  }
  double i_term = g->i[j] * *e_int;
  if (!g->antiwindup) i_term = clampd(i_term, g->i_min[j], g->i_max[j]);
  const double d_term = g->d[j] * error_dot;
  return p_term + i_term + d_term;
}

void oracle_swing_branch_leg(const oracle_swing_params *sp, const oracle_pid_params *pid, int leg, int leg_mode,
                             const double base_quat[4], const double q_id[3], const double q[3], const double qd[3],
                             const double qd_oldest[3], const double target_pos[3], const double target_vel[3],
                             const double joint_command[3], double period, double pid_error_last[3],
                             double pid_error_integral[3], double effort[3]) {
  /* base_orientation.rotate((0,0,-9.8)): third column of R times -9.8 */
  const double w = base_quat[0], x = base_quat[1], y = base_quat[2], z = base_quat[3];
  const double g[3] = {-9.8 * (2.0 * (x * z + w * y)), -9.8 * (2.0 * (y * z - w * x)), -9.8 * (1.0 - 2.0 * (x * x + y * y))};
  double G[3], tsw[3];
  oracle_leg_gravity(leg, q, g, G);
  oracle_swing_leg_torque(sp, leg, q_id, q, qd, qd_oldest, target_pos, target_vel, tsw);
  for (int k = 0; k < 3; k++) {
    const int j = 3 * leg + k;
    const double cmd = clampd(joint_command[k], pid->lower[j], pid->upper[j]);
    double e = pid_command(pid, j, cmd - q[k], period, &pid_error_last[k], &pid_error_integral[k]);
    if (leg_mode == 3 || leg_mode == 4) e = tsw[k];
    else if (leg_mode != 2) e += G[k];
    else e = G[k];
    effort[k] = e;
  }
}
