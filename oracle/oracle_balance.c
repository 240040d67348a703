/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_balance.h). */
#include "oracle_balance.h"

#include <float.h>
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle_model.h"
#include "oracle_quadprog.h"

/* ---------------------------------------------------------------- kindr --- */

/* RotationQuaternion(w,x,y,z) -> rotation matrix (Eigen toRotationMatrix form;
 * kindr stores base->world, `rotate` is the active rotation). */
void oracle_quat_to_matrix(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

static void rot(const double R[9], const double v[3], double o[3]) { /* q.rotate(v) */
  for (int i = 0; i < 3; i++) o[i] = R[i * 3] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2];
}

static void irot(const double R[9], const double v[3], double o[3]) { /* q.inverseRotate(v) */
  for (int i = 0; i < 3; i++) o[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}

static void quat_mul(const double a[4], const double b[4], double o[4]) { /* Hamilton */
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

/* a.boxMinus(b) = log(a * b^-1) as a rotation vector (kindr 1.x
 * quaternion->rotation-vector conversion: 2 acos(w)/sqrt(1-w^2) * v, with the
 * small-angle branch 2 v). */
void oracle_quat_box_minus(const double a[4], const double b[4], double out[3]) {
  const double binv[4] = {b[0], -b[1], -b[2], -b[3]};
  double d[4];
  quat_mul(a, binv, d);
  const double s2 = 1.0 - d[0] * d[0];
  if (s2 < 1e-12) {
    for (int i = 0; i < 3; i++) out[i] = 2.0 * d[1 + i];
  } else {
    const double k = 2.0 * acos(d[0]) / sqrt(s2);
    for (int i = 0; i < 3; i++) out[i] = k * d[1 + i];
  }
}

static void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

static void normalize3(double *v) {
  const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= n; v[1] /= n; v[2] /= n;
}

/* --------------------------------------------------------------- params --- */

void oracle_balance_default_params(oracle_balance_params *p) {
  /* balance_controller/config/controller_gains.yaml:1-41 */
  const double kp_t[3] = {5000, 5000, 10000}, kd_t[3] = {5000, 4000, 5000}, kff_t[3] = {10, 10, 100};
  const double kp_r[3] = {10000, 10000, 4000}, kd_r[3] = {1000, 1000, 1000}, kff_r[3] = {0.2, 0.2, 1000};
  const double S[6] = {1, 5, 1, 10, 10, 5};
  /* quadruped_state.cpp:83-97, limb order LF, RF, RH, LH */
  const double hips[4][3] = {{0.42, 0.075, 0.0}, {0.42, -0.075, 0.0}, {-0.42, -0.075, 0.0}, {-0.42, 0.075, 0.0}};
  memcpy(p->kp_trans, kp_t, sizeof(kp_t)); memcpy(p->kd_trans, kd_t, sizeof(kd_t));
  memcpy(p->kff_trans, kff_t, sizeof(kff_t));
  memcpy(p->kp_rot, kp_r, sizeof(kp_r)); memcpy(p->kd_rot, kd_r, sizeof(kd_r));
  memcpy(p->kff_rot, kff_r, sizeof(kff_r));
  memcpy(p->force_weights, S, sizeof(S));
  p->regularizer = 0.0001;
  p->friction = 0.6;
  p->min_normal_force = 10.0;
  p->torque_limit = 300.0;            /* ros_balance_controller.cpp:451-454 */
  p->torso_mass = 27.0;               /* quadruped_state.cpp:28 */
  for (int i = 0; i < 4; i++) p->leg_mass[i] = 6.0; /* quadruped_state.cpp:36-41 */
  p->gravity = 9.8;                   /* VirtualModelController.cpp:165 */
  p->grav_comp_percentage = 1.0;      /* VirtualModelController.cpp:54 */
  p->com_in_base[0] = p->com_in_base[1] = p->com_in_base[2] = 0.0; /* quadruped_state.cpp:29 */
  memcpy(p->hip_in_base, hips, sizeof(hips));
}

/* --------------------------------------------- VirtualModelController ----- */

void oracle_virtual_wrench(const oracle_balance_params *prm,
                           const double base_pos[3], const double base_quat[4],
                           const double base_linvel[3], const double base_angvel[3],
                           const double des_pos[3], const double des_quat[4],
                           const double des_linvel[3], const double des_angvel[3],
                           double wrench[6]) {
  double Rm[9];
  oracle_quat_to_matrix(base_quat, Rm);

  /* computeError, VirtualModelController.cpp:104-160 */
  double e_p[3], e_v[3], e_w[3], e_o[3];
  for (int i = 0; i < 3; i++) {
    e_p[i] = des_pos[i] - base_pos[i];
    e_v[i] = des_linvel[i] - base_linvel[i];
    e_w[i] = des_angvel[i] - base_angvel[i];
  }
  {
    /* -(q_d^-1).boxMinus(q_m^-1), :120-124 */
    const double qd_inv[4] = {des_quat[0], -des_quat[1], -des_quat[2], -des_quat[3]};
    const double qm_inv[4] = {base_quat[0], -base_quat[1], -base_quat[2], -base_quat[3]};
    oracle_quat_box_minus(qd_inv, qm_inv, e_o);
    for (int i = 0; i < 3; i++) e_o[i] = -e_o[i];
  }

  /* computeGravityCompensation, :162-188 */
  const double gW[3] = {0.0, 0.0, -prm->gravity};
  double gB[3], Fg[3], Tg[3], f_torso[3], tmp[3];
  irot(Rm, gW, gB);
  for (int i = 0; i < 3; i++) f_torso[i] = -prm->grav_comp_percentage * prm->torso_mass * gB[i];
  memcpy(Fg, f_torso, sizeof(Fg));
  cross3(prm->com_in_base, f_torso, Tg);
  for (int l = 0; l < 4; l++) {
    double f_leg[3], arm[3];
    for (int i = 0; i < 3; i++) {
      f_leg[i] = -prm->grav_comp_percentage * prm->leg_mass[l] * gB[i];
      arm[i] = prm->hip_in_base[l][i] - prm->com_in_base[i]; /* quadruped_state.cpp:83-97 */
      Fg[i] += f_leg[i];
    }
    cross3(arm, f_leg, tmp);
    for (int i = 0; i < 3; i++) Tg[i] += tmp[i];
  }

  /* computeVirtualForce, :191-239.  orientationWorldToControl is the identity
   * by construction (:204-206), so the "world frame" errors are e_p, e_v. */
  double Rep[3], Rev[3], Rff[3], fb_p[3], fb_d[3];
  const double ff_lin[3] = {des_linvel[0], des_linvel[1], 0.0};
  const double gfb[3] = {prm->kp_trans[0] * 0.0, prm->kp_trans[1] * 0.0, prm->kp_trans[2] * e_p[2]};
  const double gdb[3] = {prm->kd_trans[0] * 0.0, prm->kd_trans[1] * 0.0, prm->kd_trans[2] * e_v[2]};
  irot(Rm, e_p, Rep);
  irot(Rm, e_v, Rev);
  irot(Rm, ff_lin, Rff);
  irot(Rm, gfb, fb_p);
  irot(Rm, gdb, fb_d);
  for (int i = 0; i < 3; i++)
    wrench[i] = prm->kp_trans[i] * Rep[i] + prm->kd_trans[i] * Rev[i] + prm->kff_trans[i] * Rff[i]
              + Fg[i] + fb_p[i] + fb_d[i];

  /* computeVirtualTorque, :242-268 */
  double kd_ew[3], kff_w[3], Rd[3], Rf[3];
  for (int i = 0; i < 3; i++) kd_ew[i] = prm->kd_rot[i] * e_w[i];
  kff_w[0] = prm->kff_rot[0] * 0.0; kff_w[1] = prm->kff_rot[1] * 0.0; kff_w[2] = prm->kff_rot[2] * des_angvel[2];
  irot(Rm, kd_ew, Rd);
  irot(Rm, kff_w, Rf);
  for (int i = 0; i < 3; i++) wrench[3 + i] = prm->kp_rot[i] * e_o[i] + Rd[i] + Rf[i] + Tg[i];
}

/* ------------------------------------------- ContactForceDistribution ----- */

void oracle_force_qp_assemble(const oracle_balance_params *prm, int nS,
                              const double *r_feet, const double wrench[6],
                              const double *n_B, const double *t1, const double *t2,
                              double *G, double *g0, double *CI, double *ci0) {
  const int n = 3 * nS, m = 5 * nS;
  double A[6 * 12];
  memset(A, 0, sizeof(A));
  /* prepareOptimization, ContactForceDistribution.cpp:168-206:
   * A = [I ... I ; skew(r_1) ... skew(r_nS)] */
  for (int l = 0; l < nS; l++) {
    const double *r = r_feet + 3 * l;
    for (int i = 0; i < 3; i++) A[i * n + 3 * l + i] = 1.0;
    A[3 * n + 3 * l + 1] = -r[2]; A[3 * n + 3 * l + 2] = r[1];
    A[4 * n + 3 * l + 0] = r[2];  A[4 * n + 3 * l + 2] = -r[0];
    A[5 * n + 3 * l + 0] = -r[1]; A[5 * n + 3 * l + 1] = r[0];
  }
  /* min (Ax-b)'S(Ax-b) + x'Wx  <=>  1/2 x'(A'SA+W)x - (A'Sb)'x   (:388) */
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) {
      double acc = 0.0;
      for (int k = 0; k < 6; k++) acc += A[k * n + i] * prm->force_weights[k] * A[k * n + j];
      G[i * n + j] = acc + (i == j ? prm->regularizer : 0.0);
    }
    double acc = 0.0;
    for (int k = 0; k < 6; k++) acc += A[k * n + i] * prm->force_weights[k] * wrench[k];
    g0[i] = -acc;
  }
  /* d <= D x: rows 0..nS-1 minimal normal force (:210-252), then 4 friction
   * rows per leg (:254-336).  QuadProg++ form: CI = D', ci0 = -d. */
  memset(CI, 0, sizeof(double) * (size_t)n * (size_t)m);
  for (int l = 0; l < nS; l++) {
    const double *nb = n_B + 3 * l, *a = t1 + 3 * l, *b = t2 + 3 * l;
    for (int i = 0; i < 3; i++) {
      const int row = 3 * l + i;
      CI[row * m + l] = nb[i];
      CI[row * m + nS + 4 * l + 0] = prm->friction * nb[i] + a[i];
      CI[row * m + nS + 4 * l + 1] = prm->friction * nb[i] - a[i];
      CI[row * m + nS + 4 * l + 2] = prm->friction * nb[i] + b[i];
      CI[row * m + nS + 4 * l + 3] = prm->friction * nb[i] - b[i];
    }
    ci0[l] = -prm->min_normal_force;
    for (int k = 0; k < 4; k++) ci0[nS + 4 * l + k] = -0.0;
  }
}

int oracle_balance_step(const oracle_balance_params *prm, const double q[12],
                        const double base_pos[3], const double base_quat[4],
                        const double base_linvel[3], const double base_angvel[3],
                        const double des_pos[3], const double des_quat[4],
                        const double des_linvel[3], const double des_angvel[3],
                        const uint8_t stance[4], const double *normals_world,
                        double *tau, double *tau_raw, double *grf, double *wrench_out,
                        int *qp_iters, int *n_active) {
  double wrench[6], Rm[9];
  double t_raw[12], x_full[12];
  int status = ORACLE_QP_OK, iters = 0, nact = 0;
  memset(t_raw, 0, sizeof(t_raw));
  memset(x_full, 0, sizeof(x_full));

  oracle_virtual_wrench(prm, base_pos, base_quat, base_linvel, base_angvel,
                        des_pos, des_quat, des_linvel, des_angvel, wrench);
  oracle_quat_to_matrix(base_quat, Rm);

  /* prepareLegLoading, ContactForceDistribution.cpp:138-166 */
  int legs[4], nS = 0;
  for (int l = 0; l < 4; l++) if (stance[l]) legs[nS++] = l;

  if (nS > 0) {
    double r_feet[12], nB[12], t1[12], t2[12];
    const double ez[3] = {0, 0, 1}, ey[3] = {0, 1, 0};
    double yB[3];
    irot(Rm, ey, yB); /* orientationControlToBase.rotate(UnitY), :301-302 */
    for (int k = 0; k < nS; k++) {
      const int l = legs[k];
      oracle_leg_fk(l, q + 3 * l, r_feet + 3 * k, NULL);
      double nW[3];
      if (normals_world) memcpy(nW, normals_world + 3 * l, sizeof(nW));
      else rot(Rm, ez, nW); /* ros_balance_controller.cpp:378 */
      irot(Rm, nW, nB + 3 * k); /* :237, :286 */
      cross3(nB + 3 * k, yB, t1 + 3 * k);      normalize3(t1 + 3 * k); /* :303 */
      cross3(nB + 3 * k, t1 + 3 * k, t2 + 3 * k); normalize3(t2 + 3 * k); /* :309 */
    }
    const int n = 3 * nS, m = 5 * nS;
    double G[144], g0[12], CI[12 * 20], ci0[20], x[12], f;
    int active[21];
    oracle_force_qp_assemble(prm, nS, r_feet, wrench, nB, t1, t2, G, g0, CI, ci0);
    /* Two OOQP solves in the reference, the second pinned to the first
     * (addDesiredLegLoadConstraints :338-383) == one solve. */
    status = oracle_solve_quadprog(n, 0, m, G, g0, NULL, NULL, CI, ci0, x, &f, active, &nact, &iters);

    if (status == ORACLE_QP_OK) {
      /* computeJointTorques, :516-578 */
      const double gW[3] = {0.0, 0.0, -prm->gravity};
      double gB[3];
      irot(Rm, gW, gB);
      for (int k = 0; k < nS; k++) {
        const int l = legs[k];
        double J[9], Gq[3];
        const double fc[3] = {-x[3 * k], -x[3 * k + 1], -x[3 * k + 2]}; /* :502-503 */
        oracle_leg_jacobian(l, q + 3 * l, J);
        oracle_leg_gravity(l, q + 3 * l, gB, Gq);
        for (int j = 0; j < 3; j++) {
          t_raw[3 * l + j] = (J[0 * 3 + j] * fc[0] + J[1 * 3 + j] * fc[1] + J[2 * 3 + j] * fc[2]) + Gq[j];
          x_full[3 * l + j] = x[3 * k + j];
        }
      }
    }
  }

  for (int i = 0; i < 12; i++) {
    double t = t_raw[i];
    if (tau_raw) tau_raw[i] = t;
    if (t > prm->torque_limit) t = prm->torque_limit;   /* ros_balance_controller.cpp:451-454 */
    if (t < -prm->torque_limit) t = -prm->torque_limit;
    if (tau) tau[i] = t;
    if (grf) grf[i] = x_full[i];
  }
  if (wrench_out) memcpy(wrench_out, wrench, sizeof(wrench));
  if (qp_iters) *qp_iters = iters;
  if (n_active) *n_active = nact;
  return status;
}

void oracle_balance_batch(const oracle_balance_params *prm, int64_t B,
                          const double *q, const double *base_pos, const double *base_quat,
                          const double *base_linvel, const double *base_angvel,
                          const double *des_pos, const double *des_quat,
                          const double *des_linvel, const double *des_angvel,
                          const uint8_t *stance, const double *normals_world,
                          double *tau, double *grf, int32_t *status, int nthreads) {
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int64_t i = 0; i < B; i++) {
    int st = oracle_balance_step(prm, q + 12 * i, base_pos + 3 * i, base_quat + 4 * i,
                                 base_linvel + 3 * i, base_angvel + 3 * i, des_pos + 3 * i,
                                 des_quat + 4 * i, des_linvel + 3 * i, des_angvel + 3 * i,
                                 stance + 4 * i, normals_world ? normals_world + 12 * i : NULL,
                                 tau ? tau + 12 * i : NULL, NULL, grf ? grf + 12 * i : NULL, NULL, NULL, NULL);
    if (status) status[i] = st;
  }
  (void)nthreads;
}

/* Timing leg of bench.py's cpu_baseline: `passes` passes over the batch inside ONE parallel region (each thread
 * keeps its contiguous block of robots, no fork/join between passes), wall time taken between two barriers.
 * Returns the seconds; the outputs hold the last pass. */
double oracle_balance_batch_repeat(const oracle_balance_params *prm, int64_t B,
                                   const double *q, const double *base_pos, const double *base_quat,
                                   const double *base_linvel, const double *base_angvel,
                                   const double *des_pos, const double *des_quat,
                                   const double *des_linvel, const double *des_angvel,
                                   const uint8_t *stance, double *tau, int32_t *status, int nthreads, int passes) {
  double t0 = 0.0, t1 = 0.0;
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
  {
    const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
  {
    const int tid = 0, nt = 1;
#endif
    const int64_t lo = B * tid / nt, hi = B * (tid + 1) / nt;
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
    t0 = omp_get_wtime();
#endif
    for (int p = 0; p < passes; p++)
      for (int64_t i = lo; i < hi; i++) {
        int st = oracle_balance_step(prm, q + 12 * i, base_pos + 3 * i, base_quat + 4 * i, base_linvel + 3 * i,
                                     base_angvel + 3 * i, des_pos + 3 * i, des_quat + 4 * i, des_linvel + 3 * i,
                                     des_angvel + 3 * i, stance + 4 * i, NULL, tau + 12 * i, NULL, NULL, NULL, NULL, NULL);
        status[i] = st;
      }
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
    t1 = omp_get_wtime();
#endif
  }
  return t1 - t0;
}

/* ---- the reference's own problem statement, as handed to ooqpei::QuadraticProblemFormulation::solve ----------
 * prepareOptimization (ContactForceDistribution.cpp:168-206): b = [F ; T], S = diag(virtualForceWeights_),
 * A = [I ... I ; skew(r_1) ... skew(r_nS)], W = groundForceWeight_ I;
 * addMinimalForceConstraints (:210-252): nS rows n_B' f_i, d = minimalNormalGroundForce_, f = DBL_MAX;
 * addFrictionConstraints (:254-336): per leg (mu n + t1)', (mu n - t1)', (mu n + t2)', (mu n - t2)', d = 0, f = DBL_MAX.
 * A is [6][n], D is [5 nS][n], row-major; S and W are the diagonals. */
void oracle_force_lsq_assemble(const oracle_balance_params *prm, int nS, const double *r_feet, const double wrench[6],
                               const double *n_B, const double *t1, const double *t2, double *A, double *S, double *b,
                               double *W, double *D, double *d, double *f) {
  const int n = 3 * nS, m = 5 * nS;
  memset(A, 0, sizeof(double) * 6 * (size_t)n);
  memset(D, 0, sizeof(double) * (size_t)m * (size_t)n);
  for (int l = 0; l < nS; l++) {
    const double *r = r_feet + 3 * l;
    for (int i = 0; i < 3; i++) A[i * n + 3 * l + i] = 1.0;
    A[3 * n + 3 * l + 1] = -r[2]; A[3 * n + 3 * l + 2] = r[1];
    A[4 * n + 3 * l + 0] = r[2];  A[4 * n + 3 * l + 2] = -r[0];
    A[5 * n + 3 * l + 0] = -r[1]; A[5 * n + 3 * l + 1] = r[0];
  }
  for (int kk = 0; kk < 6; kk++) { S[kk] = prm->force_weights[kk]; b[kk] = wrench[kk]; }
  for (int i = 0; i < n; i++) W[i] = prm->regularizer;
  for (int l = 0; l < nS; l++) {
    const double *nb = n_B + 3 * l, *a = t1 + 3 * l, *c = t2 + 3 * l;
    for (int i = 0; i < 3; i++) {
      D[l * n + 3 * l + i] = nb[i];
      D[(nS + 4 * l + 0) * n + 3 * l + i] = prm->friction * nb[i] + a[i];
      D[(nS + 4 * l + 1) * n + 3 * l + i] = prm->friction * nb[i] - a[i];
      D[(nS + 4 * l + 2) * n + 3 * l + i] = prm->friction * nb[i] + c[i];
      D[(nS + 4 * l + 3) * n + 3 * l + i] = prm->friction * nb[i] - c[i];
    }
    d[l] = prm->min_normal_force;
    f[l] = DBL_MAX;                                      /* std::numeric_limits<double>::max(), :246 */
    for (int kk = 0; kk < 4; kk++) { d[nS + 4 * l + kk] = 0.0; f[nS + 4 * l + kk] = DBL_MAX; } /* :328-329 */
  }
}

/* min (Ax - b)'S(Ax - b) + x'Wx  s.t. Cx = c, d <= Dx <= f -- the contract of ooqpei::QuadraticProblemFormulation::solve
 * (third-party, absent: PARITY UNPINNED at this boundary; the problem is strictly convex, the target is its unique
 * minimiser).  Route, deliberately different from the device's (which projects the rows out of an explicit inverse one
 * by one): the equality rows are eliminated with a rank-revealing, pivoted Gram-Schmidt into x = x_p + Z y, the reduced
 * problem in y is solved by the pinned Goldfarb-Idnani restatement without equalities, bounds of +-DBL_MAX / inf dropped.
 * Returns ORACLE_QP_OK / ORACLE_QP_INFEASIBLE (inconsistent equalities or empty feasible set) / ORACLE_QP_NOT_PD. */
int oracle_weighted_lsq_qp(int n, int k, int p, int m, const double *A, const double *S, const double *b, const double *W,
                           const double *C, const double *c, const double *D, const double *d, const double *f, double *x) {
  double G[12 * 12], g0[12], Q[12][12], xp[12];
  int rank = 0;
  if (n < 1 || n > 12 || k < 1 || k > 12 || p < 0 || p > 12 || m < 0 || m > 24) return ORACLE_QP_INFEASIBLE;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) {
      double acc = 0.0;
      for (int r = 0; r < k; r++) acc += A[r * n + i] * S[r] * A[r * n + j];
      G[i * n + j] = 2.0 * (acc + (i == j ? W[i] : 0.0));
    }
    double acc = 0.0;
    for (int r = 0; r < k; r++) acc += A[r * n + i] * S[r] * b[r];
    g0[i] = -2.0 * acc;
    xp[i] = 0.0;
  }
  /* orthonormal basis Q[0..rank) of the row space of C (modified Gram-Schmidt, twice), particular solution xp */
  for (int r = 0; r < p; r++) {
    double v[12], nrm0 = 0.0, nrm = 0.0;
    for (int i = 0; i < n; i++) { v[i] = C[r * n + i]; nrm0 += v[i] * v[i]; }
    for (int pass = 0; pass < 2; pass++)
      for (int q = 0; q < rank; q++) {
        double dot = 0.0;
        for (int i = 0; i < n; i++) dot += Q[q][i] * v[i];
        for (int i = 0; i < n; i++) v[i] -= dot * Q[q][i];
      }
    for (int i = 0; i < n; i++) nrm += v[i] * v[i];
    double cx = 0.0;
    for (int i = 0; i < n; i++) cx += C[r * n + i] * xp[i];
    if (nrm0 == 0.0 || nrm <= 1e-24 * nrm0) {            /* zero row or in the span of the rows before it */
      if (fabs(c[r] - cx) > 1e-9 * (1.0 + fabs(c[r]))) return ORACLE_QP_INFEASIBLE;
      continue;
    }
    nrm = sqrt(nrm);
    for (int i = 0; i < n; i++) Q[rank][i] = v[i] / nrm;
    /* move xp along the new direction until row r holds: C_r (xp + t q) = c_r, and C_r q = |v| by construction */
    const double t = (c[r] - cx) / nrm;
    for (int i = 0; i < n; i++) xp[i] += t * Q[rank][i];
    rank++;
  }
  /* null-space basis Z: complete Q to an orthonormal basis of R^n with the unit vectors */
  double Z[12][12];
  int nz = 0;
  for (int e = 0; e < n && rank + nz < n; e++) {
    double v[12], nrm = 0.0;
    for (int i = 0; i < n; i++) v[i] = (i == e) ? 1.0 : 0.0;
    for (int pass = 0; pass < 2; pass++) {
      for (int q = 0; q < rank; q++) {
        double dot = 0.0;
        for (int i = 0; i < n; i++) dot += Q[q][i] * v[i];
        for (int i = 0; i < n; i++) v[i] -= dot * Q[q][i];
      }
      for (int q = 0; q < nz; q++) {
        double dot = 0.0;
        for (int i = 0; i < n; i++) dot += Z[q][i] * v[i];
        for (int i = 0; i < n; i++) v[i] -= dot * Z[q][i];
      }
    }
    for (int i = 0; i < n; i++) nrm += v[i] * v[i];
    if (nrm < 1e-6) continue;
    nrm = sqrt(nrm);
    for (int i = 0; i < n; i++) Z[nz][i] = v[i] / nrm;
    nz++;
  }
  /* one-sided rows CI'x + ci0 >= 0 in x, then reduced to y */
  double CIx[48][12], ci0x[48];
  int mi = 0;
  for (int r = 0; r < m; r++) {
    if (d[r] > -DBL_MAX) { for (int i = 0; i < n; i++) CIx[mi][i] = D[r * n + i]; ci0x[mi++] = -d[r]; }
    if (f[r] < DBL_MAX) { for (int i = 0; i < n; i++) CIx[mi][i] = -D[r * n + i]; ci0x[mi++] = f[r]; }
  }
  if (nz == 0) { /* the equalities pin x: feasible or not */
    for (int r = 0; r < mi; r++) {
      double s = ci0x[r];
      for (int i = 0; i < n; i++) s += CIx[r][i] * xp[i];
      if (s < -1e-9 * (1.0 + fabs(ci0x[r]))) return ORACLE_QP_INFEASIBLE;
    }
    memcpy(x, xp, sizeof(double) * (size_t)n);
    return ORACLE_QP_OK;
  }
  double Gy[12 * 12], gy[12], CIy[12 * 48], ciy[48], GZ[12][12], y[12], fval;
  for (int a = 0; a < nz; a++)
    for (int i = 0; i < n; i++) {
      double acc = 0.0;
      for (int j = 0; j < n; j++) acc += G[i * n + j] * Z[a][j];
      GZ[a][i] = acc;
    }
  for (int a = 0; a < nz; a++) {
    for (int bb = 0; bb < nz; bb++) {
      double acc = 0.0;
      for (int i = 0; i < n; i++) acc += Z[a][i] * GZ[bb][i];
      Gy[a * nz + bb] = acc;
    }
    double acc = 0.0;
    for (int i = 0; i < n; i++) {
      double gxp = g0[i];
      for (int j = 0; j < n; j++) gxp += G[i * n + j] * xp[j];
      acc += Z[a][i] * gxp;
    }
    gy[a] = acc;
  }
  for (int r = 0; r < mi; r++) {
    double s = ci0x[r];
    for (int i = 0; i < n; i++) s += CIx[r][i] * xp[i];
    ciy[r] = s;
    for (int a = 0; a < nz; a++) {
      double acc = 0.0;
      for (int i = 0; i < n; i++) acc += CIx[r][i] * Z[a][i];
      CIy[a * mi + r] = acc;
    }
  }
  int active[64], nact = 0, iters = 0;
  const int st = oracle_solve_quadprog(nz, 0, mi, Gy, gy, NULL, NULL, CIy, ciy, y, &fval, active, &nact, &iters);
  if (st != ORACLE_QP_OK) return st;
  for (int i = 0; i < n; i++) {
    double acc = xp[i];
    for (int a = 0; a < nz; a++) acc += Z[a][i] * y[a];
    x[i] = acc;
  }
  return ORACLE_QP_OK;
}
