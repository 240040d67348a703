/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into the product library.
 *
 * CPU restatement of the reference's pose optimisation (BASELINE config 5):
 *   objective   free_gait_core/src/pose_optimization/PoseOptimizationObjectiveFunction.cpp:62-248
 *   constraints free_gait_core/src/pose_optimization/PoseOptimizationFunctionConstraints.cpp:95-194
 *   plus        free_gait_core/src/pose_optimization/poseparameterization.cpp:37-51
 *   SQP loop    qp_solver/src/sequencequadraticproblemsolver.cpp:18-102
 *   QP wrapper  qp_solver/src/quadraticproblemsolver.cpp:65-97,133-207  (CI = -A', dummy zero equality)
 *   driver      free_gait_core/src/pose_optimization/PoseOptimizationSQP.cpp:58-111 (tol 0.05, 30 iterations)
 * The inner QP is oracle_quadprog.c (pinned against the reference's compiled QuadProg++),
 * including the all-zero equality column the reference always passes.
 *
 * Third-party pieces restated from upstream, UNPINNED (absent here, no reference test pins them):
 *   grid_map::Polygon::getCentroid / convertToInequalityConstraints (grid_map_core),
 *   kindr quaternion exp map / boxPlus.
 * Pinning that exists: the reference's own (unbuilt) known-answer tests
 * free_gait_core/test/PoseOptimizationSQPTest.cpp:39-199, restated in tests/test_pose_sqp_host.py.
 */
#include "oracle_pose_sqp.h"

#include <math.h>
#include <string.h>

#include "oracle_quadprog.h"

static void quat_to_mat(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
static void mv(const double R[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; i++) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
static void mtv(const double R[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; i++) o[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}
static void skew(const double r[3], double S[9]) { /* kindr::getSkewMatrixFromVector */
  S[0] = 0;     S[1] = -r[2]; S[2] = r[1];
  S[3] = r[2];  S[4] = 0;     S[5] = -r[0];
  S[6] = -r[1]; S[7] = r[0];  S[8] = 0;
}
static void mm(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

/* grid_map::Polygon::getCentroid (shoelace), vertices in the given order */
void oracle_polygon_centroid(int nv, const double *v, double c[2]) {
  double area = 0.0;
  c[0] = c[1] = 0.0;
  for (int i = 0; i < nv; i++) {
    const double *a = v + 2 * i, *b = v + 2 * ((i + 1) % nv);
    const double cr = a[0] * b[1] - b[0] * a[1];
    area += cr;
    c[0] += cr * (a[0] + b[0]);
    c[1] += cr * (a[1] + b[1]);
  }
  area *= 0.5;
  c[0] /= (6.0 * area);
  c[1] /= (6.0 * area);
}

/* grid_map::Polygon::convertToInequalityConstraints: for each edge (v_i, v_{i+1}) of the polygon
 * centred at the vertex mean c, the row a solves a.(v_i - c) = 1, a.(v_{i+1} - c) = 1; then
 * A x <= 1 + A c.  Degenerate edges (rank < 2) are skipped.  Returns the number of rows. */
int oracle_polygon_halfspaces(int nv, const double *v, double *A, double *b) {
  double c[2] = {0, 0};
  for (int i = 0; i < nv; i++) { c[0] += v[2 * i]; c[1] += v[2 * i + 1]; }
  c[0] /= nv; c[1] /= nv;
  int rows = 0;
  for (int i = 0; i < nv; i++) {
    const double x1 = v[2 * i] - c[0], y1 = v[2 * i + 1] - c[1];
    const double x2 = v[2 * ((i + 1) % nv)] - c[0], y2 = v[2 * ((i + 1) % nv) + 1] - c[1];
    const double det = x1 * y2 - x2 * y1;
    if (fabs(det) <= 1e-12 * (fabs(x1 * y2) + fabs(x2 * y1) + 1e-300)) continue;
    const double a0 = (y2 - y1) / det, a1 = (x1 - x2) / det;
    A[2 * rows] = a0; A[2 * rows + 1] = a1;
    b[rows] = 1.0 + (a0 * c[0] + a1 * c[1]);
    rows++;
  }
  return rows;
}

/* q.boxPlus(d) = exp(d) * q  (poseparameterization.cpp:42-49; kindr exp map) */
void oracle_quat_box_plus(const double q[4], const double d[3], double out[4]) {
  const double v = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double e[4];
  if (v < 1e-12) {
    e[0] = 1.0; e[1] = 0.5 * d[0]; e[2] = 0.5 * d[1]; e[3] = 0.5 * d[2];
  } else {
    const double a = 0.5 * v, s = sin(a) / v;
    e[0] = cos(a); e[1] = s * d[0]; e[2] = s * d[1]; e[3] = s * d[2];
  }
  out[0] = e[0] * q[0] - e[1] * q[1] - e[2] * q[2] - e[3] * q[3];
  out[1] = e[0] * q[1] + e[1] * q[0] + e[2] * q[3] - e[3] * q[2];
  out[2] = e[0] * q[2] - e[1] * q[3] + e[2] * q[0] + e[3] * q[1];
  out[3] = e[0] * q[3] + e[1] * q[2] - e[2] * q[1] + e[3] * q[0];
}

/* PoseOptimizationObjectiveFunction::computeValue, :62-101 */
double oracle_pose_cost(const oracle_pose_problem *pb, const double pose[7]) {
  double R[9], value = 0.0, c2[2];
  quat_to_mat(pose + 3, R);
  for (int k = 0; k < pb->n_legs; k++) {
    const int l = pb->leg_order[k];
    double Pd[3];
    mv(R, pb->nominal[l], Pd);
    for (int i = 0; i < 3; i++) {
      const double e = pose[i] + Pd[i] - pb->stance[l][i];
      value += e * e;
    }
  }
  double Pr[3];
  mv(R, pb->r_com, Pr);
  oracle_polygon_centroid(pb->n_vertices, &pb->polygon[0][0], c2);
  const double ex = c2[0] - (pose[0] + Pr[0]), ey = c2[1] - (pose[1] + Pr[1]);
  value += pb->com_weight * (ex * ex + ey * ey);
  return value;
}

/* gradient :150-194, Hessian :196-248 (both in the local 6-vector parameterisation) */
void oracle_pose_grad_hess(const oracle_pose_problem *pb, const double pose[7], double g[6], double H[36]) {
  double R[9], ps[9];
  const double *p = pose;
  quat_to_mat(pose + 3, R);
  skew(p, ps);
  memset(g, 0, 6 * sizeof(double));
  memset(H, 0, 36 * sizeof(double));
  for (int k = 0; k < pb->n_legs; k++) {
    const int l = pb->leg_order[k];
    const double *f = pb->stance[l];
    double Pd[3], D[9], F[9], Dp[3], Df[3], T1[9], T2[9], T3[9], T4[9];
    mv(R, pb->nominal[l], Pd);
    skew(Pd, D);
    skew(f, F);
    mv(D, p, Dp);
    mv(D, f, Df);
    for (int i = 0; i < 3; i++) {
      g[i] += p[i] + Pd[i] - f[i];
      g[3 + i] += Dp[i] - Df[i];
    }
    mm(ps, D, T1); mm(D, ps, T2); mm(F, D, T3); mm(D, F, T4);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        H[6 * i + j] += (i == j) ? 1.0 : 0.0;
        H[6 * i + 3 + j] += -D[3 * i + j];
        H[6 * (3 + i) + j] += D[3 * i + j];
        H[6 * (3 + i) + 3 + j] += 0.5 * (T1[3 * i + j] + T2[3 * i + j] - T3[3 * i + j] - T4[3 * i + j]);
      }
  }
  {
    const double w = pb->com_weight;
    const double pbar[3] = {p[0], p[1], 0.0};
    double Pr[3], Rr[9], C[9], c2[2], a[3], b[3], T1[9], T2[9], T3[9], T4[9];
    mv(R, pb->r_com, Pr);
    Pr[2] = 0.0;
    skew(Pr, Rr);
    oracle_polygon_centroid(pb->n_vertices, &pb->polygon[0][0], c2);
    const double rc[3] = {c2[0], c2[1], 0.0};
    skew(rc, C);
    mv(Rr, pbar, a);
    mv(Rr, rc, b);
    for (int i = 0; i < 3; i++) {
      g[i] += w * (pbar[i] - rc[i] + Pr[i]);
      g[3 + i] += w * (a[i] - b[i]);
    }
    mm(ps, Rr, T1); mm(Rr, ps, T2); mm(C, Rr, T3); mm(Rr, C, T4);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        H[6 * i + j] += w * ((i == j && i < 2) ? 1.0 : 0.0);
        H[6 * i + 3 + j] += -w * Rr[3 * i + j];
        H[6 * (3 + i) + j] += w * Rr[3 * i + j];
        H[6 * (3 + i) + 3 + j] += 0.5 * w * (T1[3 * i + j] + T2[3 * i + j] - T3[3 * i + j] - T4[3 * i + j]);
      }
  }
  for (int i = 0; i < 6; i++) g[i] *= 2.0;
  for (int i = 0; i < 36; i++) H[i] *= 2.0;
}

/* constraint values, maxima and local Jacobian; rows: polygon half-spaces, then leg lengths
 * (PoseOptimizationFunctionConstraints.cpp:95-194).  Returns the number of rows m. */
int oracle_pose_constraints(const oracle_pose_problem *pb, const double pose[7], double *val, double *vmax,
                            double *A /* m x 6 */) {
  double R[9], GA[8], gb[4];
  const double *p = pose;
  quat_to_mat(pose + 3, R);
  const int nsp = oracle_polygon_halfspaces(pb->n_vertices, &pb->polygon[0][0], GA, gb);
  double Pr[3], Rr[9];
  mv(R, pb->r_com, Pr);
  skew(Pr, Rr);
  const double cw[2] = {p[0] + Pr[0], p[1] + Pr[1]};
  for (int i = 0; i < nsp; i++) {
    val[i] = GA[2 * i] * cw[0] + GA[2 * i + 1] * cw[1];
    vmax[i] = gb[i];
    const double G3[3] = {GA[2 * i], GA[2 * i + 1], 0.0};
    for (int j = 0; j < 3; j++) {
      A[6 * i + j] = G3[j];
      A[6 * i + 3 + j] = -(G3[0] * Rr[j] + G3[1] * Rr[3 + j] + G3[2] * Rr[6 + j]);
    }
  }
  for (int k = 0; k < pb->n_legs; k++) {
    const int l = pb->leg_order[k], row = nsp + k;
    const double *f = pb->stance[l];
    const double df[3] = {f[0] - p[0], f[1] - p[1], f[2] - p[2]};
    double bf[3], Ph[3], Hs[9];
    mtv(R, df, bf);
    const double e[3] = {bf[0] - pb->hips[l][0], bf[1] - pb->hips[l][1], bf[2] - pb->hips[l][2]};
    val[row] = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    vmax[row] = pb->max_len[l];
    mv(R, pb->hips[l], Ph);
    skew(Ph, Hs);
    double ln[3] = {p[0] + Ph[0] - f[0], p[1] + Ph[1] - f[1], p[2] + Ph[2] - f[2]};
    const double nn = sqrt(ln[0] * ln[0] + ln[1] * ln[1] + ln[2] * ln[2]);
    ln[0] /= nn; ln[1] /= nn; ln[2] /= nn;
    for (int j = 0; j < 3; j++) {
      A[6 * row + j] = ln[j];
      A[6 * row + 3 + j] = -(ln[0] * Hs[j] + ln[1] * Hs[3 + j] + ln[2] * Hs[6 + j]);
    }
  }
  return nsp + pb->n_legs;
}

int oracle_pose_sqp(const oracle_pose_problem *pb, const double pose_in[7], double tol, int max_iter,
                    int dummy_equality, double pose_out[7], int *iters_out, double *cost_out, double *dp_hist) {
  double pose[7];
  memcpy(pose, pose_in, sizeof(pose));
  int k = 0, status = ORACLE_QP_OK;
  while (k < max_iter) {
    double g[6], H[36], val[8], vmax[8], A[48], CI[48], ci0[8], CE[6] = {0, 0, 0, 0, 0, 0}, ce0[1] = {0}, dp[6], f;
    oracle_pose_grad_hess(pb, pose, g, H);
    const int m = oracle_pose_constraints(pb, pose, val, vmax, A);
    /* A dp <= vmax - val  ->  CI = -A', ci0 = b  (quadraticproblemsolver.cpp:164) */
    for (int i = 0; i < m; i++) {
      ci0[i] = vmax[i] - val[i];
      for (int j = 0; j < 6; j++) CI[j * m + i] = -A[6 * i + j];
    }
    k++;
    status = oracle_solve_quadprog(6, dummy_equality ? 1 : 0, m, H, g, CE, ce0, CI, ci0, dp, &f, NULL, NULL, NULL);
    if (status != ORACLE_QP_OK) break;
    if (dp_hist) memcpy(dp_hist + 6 * (k - 1), dp, sizeof(dp));
    for (int i = 0; i < 3; i++) pose[i] += dp[i];
    double qn[4];
    oracle_quat_box_plus(pose + 3, dp + 3, qn);
    memcpy(pose + 3, qn, sizeof(qn));
    const double nrm = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2] + dp[3] * dp[3] + dp[4] * dp[4] + dp[5] * dp[5]);
    if (nrm < tol) break; /* sequencequadraticproblemsolver.cpp:72-76 */
  }
  memcpy(pose_out, pose, sizeof(pose));
  if (iters_out) *iters_out = k;
  if (cost_out) *cost_out = oracle_pose_cost(pb, pose);
  return status;
}

/* ------------------------------------------------------------------ PoseOptimizationQP, PoseConstraintsChecker */

int oracle_pose_qp(const oracle_pose_problem *pb, const double pose_in[7], int dummy_equality, double pose_out[7]) {
  double R[9], G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, q[3] = {0, 0, 0};
  quat_to_mat(pose_in + 3, R);
  /* P = 2 A'A = 2 nFeet I;  q = -2 A'b,  b_i = f_i - R d_i  (:57-66, :83-84) */
  for (int k = 0; k < pb->n_legs; k++) {
    const int l = pb->leg_order[k];
    double Rd[3];
    mv(R, pb->nominal[l], Rd);
    for (int i = 0; i < 3; i++) q[i] += -2.0 * (pb->stance[l][i] - Rd[i]);
  }
  for (int i = 0; i < 3; i++) G[4 * i] = 2.0 * pb->n_legs;
  /* G x <= h,  h = hp - G (R r_com)_xy;  z column zero  (:73-81) */
  double GA[8], gb[4], Rr[3], CI[12], ci0[4], CE[3] = {0, 0, 0}, ce0[1] = {0}, x[3], f;
  const int m = oracle_polygon_halfspaces(pb->n_vertices, &pb->polygon[0][0], GA, gb);
  mv(R, pb->r_com, Rr);
  for (int i = 0; i < m; i++) {
    ci0[i] = gb[i] - (GA[2 * i] * Rr[0] + GA[2 * i + 1] * Rr[1]);
    CI[0 * m + i] = -GA[2 * i];
    CI[1 * m + i] = -GA[2 * i + 1];
    CI[2 * m + i] = -0.0;
  }
  const int st = oracle_solve_quadprog(3, dummy_equality ? 1 : 0, m, G, q, CE, ce0, CI, ci0, x, &f, NULL, NULL, NULL);
  memcpy(pose_out, pose_in, 7 * sizeof(double));
  if (st == ORACLE_QP_OK) memcpy(pose_out, x, sizeof(x));
  return st;
}

/* grid_map::Polygon::isInside (crossing number) */
int oracle_polygon_is_inside(int nv, const double *v, const double pt[2]) {
  int cross = 0;
  for (int i = 0, j = nv - 1; i < nv; j = i++) {
    const double xi = v[2 * i], yi = v[2 * i + 1], xj = v[2 * j], yj = v[2 * j + 1];
    if (((yi > pt[1]) != (yj > pt[1])) && (pt[0] < (xj - xi) * (pt[1] - yi) / (yj - yi) + xi)) cross++;
  }
  return cross % 2;
}

int oracle_pose_check(const oracle_pose_problem *pb, const double pose[7], const double min_len[4], double leg_tol) {
  double R[9], Pr[3];
  quat_to_mat(pose + 3, R);
  mv(R, pb->r_com, Pr);
  const double com[2] = {pose[0] + Pr[0], pose[1] + Pr[1]};
  if (!oracle_polygon_is_inside(pb->n_vertices, &pb->polygon[0][0], com)) return 0;
  for (int k = 0; k < pb->n_legs; k++) {
    const int l = pb->leg_order[k];
    const double df[3] = {pb->stance[l][0] - pose[0], pb->stance[l][1] - pose[1], pb->stance[l][2] - pose[2]};
    double bf[3];
    mtv(R, df, bf);
    const double e[3] = {bf[0] - pb->hips[l][0], bf[1] - pb->hips[l][1], bf[2] - pb->hips[l][2]};
    const double len = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    if (len < min_len[l] - leg_tol || len > pb->max_len[l] + leg_tol) return 0;
  }
  return 1;
}

void oracle_pose_sqp_batch(const oracle_pose_problem *pbs, const double *pose_in, long long B, double tol, int max_iter,
                           int dummy_equality, double *pose_out, int *iters, int *status, int nthreads) {
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (long long i = 0; i < B; i++) {
    int it = 0;
    double cost = 0.0;
    const int st = oracle_pose_sqp(&pbs[i], pose_in + 7 * i, tol, max_iter, dummy_equality, pose_out + 7 * i, &it, &cost, NULL);
    if (iters) iters[i] = it;
    if (status) status[i] = st;
  }
  (void)nthreads;
}
