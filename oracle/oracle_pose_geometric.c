/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into the product library.
 *
 * CPU restatement of free_gait::PoseOptimizationGeometric::optimize
 * (free_gait_core/src/pose_optimization/PoseOptimizationGeometric.cpp:34-105) and of the sequence
 * BaseAuto::optimizePose runs (free_gait_core/src/base_motion/BaseAuto.cpp:394-400):
 * geometric -> QP -> constraints check -> SQP when the check fails.
 *
 * PARITY UNPINNED: kindr (quaternion matrices, setUnique, rotation-vector maps), Eigen
 * (EigenSolver, setFromTwoVectors) and grid_map (centroid) are absent and the reference has no test
 * for this class.  What pins the restatement instead: the 4x4 eigen-problem is Horn's closed form of
 * the orthogonal Procrustes problem (Bloesch 2016, eq. 38-46, cited at :51-60), so the orientation
 * before the heading/roll-pitch post-processing must equal the SVD (Kabsch) solution -- checked in
 * tests/test_pose_geometric.py.
 */
#include <math.h>
#include <string.h>

#include "oracle_pose_sqp.h"

void oracle_quat_box_minus(const double a[4], const double b[4], double out[3]); /* oracle_balance.c */

static void quat_mul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

/* kindr Quaternion::getQuaternionMatrix (left product, q (x) p = Q(q) p) of the pure quaternion
 * (0, a) minus getConjugateQuaternionMatrix (right product, p (x) q = Qc(q) p) of (0, b): :63-65 */
static void ak_matrix(const double a[3], const double b[3], double A[16]) {
  const double L[16] = {0, -a[0], -a[1], -a[2], a[0], 0, -a[2], a[1], a[1], a[2], 0, -a[0], a[2], -a[1], a[0], 0};
  const double R[16] = {0, -b[0], -b[1], -b[2], b[0], 0, b[2], -b[1], b[1], -b[2], 0, b[0], b[2], b[1], -b[0], 0};
  for (int i = 0; i < 16; i++) A[i] = L[i] - R[i];
}

static void mm4(const double A[16], const double B[16], double C[16]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0.0;
      for (int k = 0; k < 4; k++) s += A[4 * i + k] * B[4 * k + j];
      C[4 * i + j] = s;
    }
}

/* Cyclic Jacobi on the symmetric part of C; eigenvalues in w, eigenvectors in the columns of V.
 * (The reference calls Eigen::EigenSolver, :69; any convergent symmetric eigen-solver returns the same
 * dominant eigenvector up to sign, which setUnique removes.) */
void oracle_sym4_eigen(const double C[16], double w[4], double V[16]) {
  double A[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      A[4 * i + j] = 0.5 * (C[4 * i + j] + C[4 * j + i]);
      V[4 * i + j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 12; sweep++) {
    /* converged when the off-diagonal mass is below 1e-20 of the diagonal's (quadratic convergence: 4-6 sweeps) */
    double off = 0.0, dia = 0.0;
    for (int p = 0; p < 4; p++) {
      dia += fabs(A[5 * p]);
      for (int q = p + 1; q < 4; q++) off += fabs(A[4 * p + q]);
    }
    if (off <= 1e-20 * dia) break;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        const double apq = A[4 * p + q];
        if (apq == 0.0) continue;
        const double theta = (A[4 * q + q] - A[4 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; k++) { /* columns p, q of A and V */
          const double akp = A[4 * k + p], akq = A[4 * k + q];
          A[4 * k + p] = c * akp - s * akq;
          A[4 * k + q] = s * akp + c * akq;
          const double vkp = V[4 * k + p], vkq = V[4 * k + q];
          V[4 * k + p] = c * vkp - s * vkq;
          V[4 * k + q] = s * vkp + c * vkq;
        }
        for (int k = 0; k < 4; k++) { /* rows p, q of A */
          const double apk = A[4 * p + k], aqk = A[4 * q + k];
          A[4 * p + k] = c * apk - s * aqk;
          A[4 * q + k] = s * apk + c * aqk;
        }
      }
  }
  for (int i = 0; i < 4; i++) w[i] = A[5 * i];
}

/* kindr RotationQuaternion::setUnique: first non-zero of (w, x, y, z) made positive */
static void quat_set_unique(double q[4]) {
  for (int i = 0; i < 4; i++) {
    if (q[i] > 0.0) return;
    if (q[i] < 0.0) { for (int k = 0; k < 4; k++) q[k] = -q[k]; return; }
  }
}

/* Eigen Quaternion::setFromTwoVectors(a, b) (kindr setFromVectors): rotation taking a to b.  In the
 * antiparallel case Eigen takes the axis from an SVD null space (any unit vector normal to a); the
 * heading vectors here lie in the xy plane, so that axis is z. */
static void quat_from_two_vectors(const double a[3], const double b[3], double q[4]) {
  const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  const double v0[3] = {a[0] / na, a[1] / na, a[2] / na}, v1[3] = {b[0] / nb, b[1] / nb, b[2] / nb};
  double c = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
  if (c < -1.0 + 1e-12) {
    c = c > -1.0 ? c : -1.0;
    const double w2 = (1.0 + c) * 0.5, k = sqrt(1.0 - w2);
    q[0] = sqrt(w2); q[1] = 0.0; q[2] = 0.0; q[3] = k;
    return;
  }
  const double s = sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
  q[0] = s * 0.5;
  q[1] = (v0[1] * v1[2] - v0[2] * v1[1]) * invs;
  q[2] = (v0[2] * v1[0] - v0[0] * v1[2]) * invs;
  q[3] = (v0[0] * v1[1] - v0[1] * v1[0]) * invs;
}

/* pose_out = (x, y, z, qw, qx, qy, qz).  stance_for_orientation is indexed by limb id
 * (LF, RF, RH, LH); q_procrustes (may be NULL) receives the orientation before the heading step. */
int oracle_pose_geometric(const oracle_pose_problem *pb, const double stance_for_orientation[4][3], double pose_out[7],
                          double q_procrustes[4]) {
  /* position: centroid of the support region, mean height offset (:38-47) */
  double cen[2];
  oracle_polygon_centroid(pb->n_vertices, &pb->polygon[0][0], cen);
  double z = 0.0;
  for (int k = 0; k < pb->n_legs; k++) {
    const int l = pb->leg_order[k];
    z += pb->stance[l][2] - pb->nominal[l][2];
  }
  z /= (double)pb->n_legs;
  pose_out[0] = cen[0]; pose_out[1] = cen[1]; pose_out[2] = z;

  /* orientation: C = sum Ak^2 - n Abar^2, eigenvector of the largest eigenvalue (:61-73) */
  double Cm[16], Am[16], Ak[16], Ak2[16];
  memset(Cm, 0, sizeof(Cm));
  memset(Am, 0, sizeof(Am));
  for (int k = 0; k < pb->n_legs; k++) {
    const int l = pb->leg_order[k];
    ak_matrix(pb->stance[l], pb->nominal[l], Ak);
    mm4(Ak, Ak, Ak2);
    for (int i = 0; i < 16; i++) { Cm[i] += Ak2[i]; Am[i] += Ak[i]; }
  }
  for (int i = 0; i < 16; i++) Am[i] = Am[i] / (double)pb->n_legs;
  mm4(Am, Am, Ak2);
  for (int i = 0; i < 16; i++) Cm[i] -= (double)pb->n_legs * Ak2[i];
  double w[4], V[16];
  oracle_sym4_eigen(Cm, w, V);
  int best = 0;
  for (int i = 1; i < 4; i++)
    if (w[i] > w[best]) best = i;
  double q[4] = {V[best], V[4 + best], V[8 + best], V[12 + best]};
  const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= nq;
  quat_set_unique(q);
  if (q_procrustes) memcpy(q_procrustes, q, 4 * sizeof(double));

  /* heading from the fore / hind mid points (:76-81); limb ids LF, RF, RH, LH = 0..3 */
  double dir[3];
  for (int i = 0; i < 3; i++)
    dir[i] = 0.5 * (stance_for_orientation[0][i] + stance_for_orientation[1][i]) -
             0.5 * (stance_for_orientation[3][i] + stance_for_orientation[2][i]);
  dir[2] = 0.0;
  const double ex[3] = {1.0, 0.0, 0.0};
  double heading[4];
  quat_from_two_vectors(ex, dir, heading);

  /* roll/pitch kept at 70 % (:84-87): yaw = exp(z part of log q), rollPitch = exp(0.7 log(yaw^-1 q)) */
  const double ident[4] = {1.0, 0.0, 0.0, 0.0};
  double rv[3], yaw[4], rel[4], rp[4];
  oracle_quat_box_minus(q, ident, rv);
  const double rvz[3] = {0.0, 0.0, rv[2]};
  oracle_quat_box_plus(ident, rvz, yaw);
  const double yaw_inv[4] = {yaw[0], -yaw[1], -yaw[2], -yaw[3]};
  quat_mul(yaw_inv, q, rel);
  oracle_quat_box_minus(rel, ident, rv);
  for (int i = 0; i < 3; i++) rv[i] *= 0.7;
  oracle_quat_box_plus(ident, rv, rp);
  quat_mul(heading, rp, pose_out + 3);
  return 0;
}

/* BaseAuto::optimizePose (BaseAuto.cpp:394-400).  stage_out: 2 = QP result accepted by the checker,
 * 3 = SQP ran.  Returns 0 on success, the failing QP status otherwise (pose_out then holds the last
 * pose reached, as the reference leaves it). */
int oracle_base_auto_optimize_pose(const oracle_pose_problem *pb, const double stance_for_orientation[4][3],
                                   const double min_len[4], double leg_tol, double sqp_tol, int sqp_max_iter,
                                   int dummy_equality, double pose_out[7], int *stage_out) {
  double pose[7], next[7];
  oracle_pose_geometric(pb, stance_for_orientation, pose, NULL);     /* result ignored (:396) */
  int st = oracle_pose_qp(pb, pose, dummy_equality, next);
  memcpy(pose_out, next, sizeof(next));
  *stage_out = 2;
  if (st != 0) return st;
  if (oracle_pose_check(pb, next, min_len, leg_tol)) return 0;
  *stage_out = 3;
  int iters = 0;
  double cost = 0.0;
  st = oracle_pose_sqp(pb, next, sqp_tol, sqp_max_iter, dummy_equality, pose_out, &iters, &cost, NULL);
  return st;
}
