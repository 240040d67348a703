// TEST INFRASTRUCTURE (host mirror): one-problem-at-a-time drivers of the pose optimisation around the step-by-step
// restatement of quadprogpp::solve_quadprog in gi_core.hpp / gi6_core.hpp.  They were the library's first kernels
// (one lane per problem); the library now runs the lane-cooperative forms of csrc/pose_coop.hpp and csrc/qp_coop.hpp
// only, and these stay as an independent second implementation that the CPU-only tests compare with the oracle
// (their host build is bit-identical to it) and the GPU tests compare the kernels with.
#pragma once

#include "gi6_core.hpp"
#include "gi_core.hpp"
#include "pose_core.hpp"

namespace qlamd {

typedef GiLayout<6, 1, 8> PoseGi;

// Linearise at `pose`: Hessian / gradient through `put_G(i, v)` / `put_g0(i, v)`, constraints into
// the scratch at offsets (kCI, kCI0): CI = -A' (stride m), CI0 = vmax - val.  Returns m.
template <int kCI, int kCI0, class Scr, class PutG, class PutG0>
QL_HD int pose_linearise_to(const PoseParamsDev &P, const PoseProblem &pb, const double centroid[2], int nsp,
                            const double GA[4][2], const double gb[4], const double pose[7], Scr &s, PutG put_G,
                            PutG0 put_g0) {
  struct Ly { enum { CI = kCI, CI0 = kCI0 }; };
  double R[9], ps[9];
  const double *p = pose;
  quat_to_matrix(pose + 3, R);
  skew3(p, ps);
  double g[6] = {0, 0, 0, 0, 0, 0}, H[36];
  for (int i = 0; i < 36; i++) H[i] = 0.0;
  int nl = 0;
  QL_UNROLL for (int k = 0; k < 4; k++) {
    if (!((pb.present >> k) & 1u)) continue;
    nl++;
    const double *f = pb.stance[k];
    double Pd[3], D[9], F[9], Dp[3], Df[3], T1[9], T2[9], T3[9], T4[9];
    rot(R, pb.nominal[k], Pd);
    skew3(Pd, D); skew3(f, F);
    rot(D, p, Dp); rot(D, f, Df);
    for (int i = 0; i < 3; i++) { g[i] += p[i] + Pd[i] - f[i]; g[3 + i] += Dp[i] - Df[i]; }
    mm3(ps, D, T1); mm3(D, ps, T2); mm3(F, D, T3); mm3(D, F, T4);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        H[6 * i + j] += (i == j) ? 1.0 : 0.0;
        H[6 * i + 3 + j] += -D[3 * i + j];
        H[6 * (3 + i) + j] += D[3 * i + j];
        H[6 * (3 + i) + 3 + j] += 0.5 * (T1[3 * i + j] + T2[3 * i + j] - T3[3 * i + j] - T4[3 * i + j]);
      }
  }
  double Pr3[3], Rr3[9]; // full Phi r_com (constraints use it with its z component)
  rot(R, pb.r_com, Pr3);
  skew3(Pr3, Rr3);
  {
    const double w = P.com_weight;
    const double pbar[3] = {p[0], p[1], 0.0};
    const double Pr[3] = {Pr3[0], Pr3[1], 0.0};
    const double rc[3] = {centroid[0], centroid[1], 0.0};
    double Rr[9], C[9], a[3], b[3], T1[9], T2[9], T3[9], T4[9];
    skew3(Pr, Rr); skew3(rc, C);
    rot(Rr, pbar, a); rot(Rr, rc, b);
    for (int i = 0; i < 3; i++) { g[i] += w * (pbar[i] - rc[i] + Pr[i]); g[3 + i] += w * (a[i] - b[i]); }
    mm3(ps, Rr, T1); mm3(Rr, ps, T2); mm3(C, Rr, T3); mm3(Rr, C, T4);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        H[6 * i + j] += w * ((i == j && i < 2) ? 1.0 : 0.0);
        H[6 * i + 3 + j] += -w * Rr[3 * i + j];
        H[6 * (3 + i) + j] += w * Rr[3 * i + j];
        H[6 * (3 + i) + 3 + j] += 0.5 * w * (T1[3 * i + j] + T2[3 * i + j] - T3[3 * i + j] - T4[3 * i + j]);
      }
  }
  for (int i = 0; i < 6; i++) put_g0(i, 2.0 * g[i]);
  for (int i = 0; i < 36; i++) put_G(i, 2.0 * H[i]);

  const int m = nsp + nl;
  const double cw[2] = {p[0] + Pr3[0], p[1] + Pr3[1]};
  QL_UNROLL for (int i = 0; i < 4; i++) {
    if (i >= nsp) continue;
    const double val = GA[i][0] * cw[0] + GA[i][1] * cw[1];
    s.at(Ly::CI0 + i) = gb[i] - val;
    const double G3[3] = {GA[i][0], GA[i][1], 0.0};
    QL_UNROLL for (int j = 0; j < 3; j++) {
      s.at(Ly::CI + j * m + i) = -G3[j];
      s.at(Ly::CI + (3 + j) * m + i) = (G3[0] * Rr3[j] + G3[1] * Rr3[3 + j] + G3[2] * Rr3[6 + j]);
    }
  }
  int row = nsp;
  QL_UNROLL for (int k = 0; k < 4; k++) {
    if (!((pb.present >> k) & 1u)) continue;
    const double *f = pb.stance[k];
    const double df[3] = {f[0] - p[0], f[1] - p[1], f[2] - p[2]};
    double bf[3], Ph[3], Hs[9];
    irot(R, df, bf);
    const double e[3] = {bf[0] - pb.hips[k][0], bf[1] - pb.hips[k][1], bf[2] - pb.hips[k][2]};
    const double len = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    s.at(Ly::CI0 + row) = pb.max_len[k] - len;
    rot(R, pb.hips[k], Ph);
    skew3(Ph, Hs);
    double ln[3] = {p[0] + Ph[0] - f[0], p[1] + Ph[1] - f[1], p[2] + Ph[2] - f[2]};
    const double nn = sqrt(ln[0] * ln[0] + ln[1] * ln[1] + ln[2] * ln[2]);
    ln[0] /= nn; ln[1] /= nn; ln[2] /= nn;
    for (int j = 0; j < 3; j++) {
      s.at(Ly::CI + j * m + row) = -ln[j];
      s.at(Ly::CI + (3 + j) * m + row) = (ln[0] * Hs[j] + ln[1] * Hs[3 + j] + ln[2] * Hs[6 + j]);
    }
    row++;
  }
  return m;
}

// LDS-resident variant (gi_core.hpp layout): everything into the scratch, CE = 0, CE0 = 0.
template <class Scr>
QL_HD int pose_linearise(const PoseParamsDev &P, const PoseProblem &pb, const double centroid[2], int nsp,
                         const double GA[4][2], const double gb[4], const double pose[7], Scr &s) {
  const int m = pose_linearise_to<PoseGi::CI, PoseGi::CI0>(
      P, pb, centroid, nsp, GA, gb, pose, s, [&](int i, double v) { s.at(PoseGi::G + i) = v; },
      [&](int i, double v) { s.at(PoseGi::G0 + i) = v; });
  for (int j = 0; j < 6; j++) s.at(PoseGi::CE + j) = 0.0;
  s.at(PoseGi::CE0) = 0.0;
  return m;
}

// The SQP loop of sequencequadraticproblemsolver.cpp:18-102.  pose is updated in place.
template <class Scr>
QL_HD int pose_sqp(const PoseParamsDev &P, const PoseProblem &pb, Scr &s, double pose[7], int *iters_out) {
  double centroid[2], GA[4][2], gb[4];
  polygon_centroid(pb.n_vertices, pb.polygon, centroid);
  const int nsp = polygon_halfspaces(pb.n_vertices, pb.polygon, GA, gb);
  int k = 0, status = kStatusOk;
  while (k < P.max_iter) {
    const int m = pose_linearise(P, pb, centroid, nsp, GA, gb, pose, s);
    k++;
    double f;
    status = gi_solve<6, 1, 8>(s, 6, P.dummy_equality ? 1 : 0, m, &f, nullptr);
    if (status != kStatusOk) break;
    double dp[6];
    for (int i = 0; i < 6; i++) dp[i] = s.at(PoseGi::X + i);
    for (int i = 0; i < 3; i++) pose[i] += dp[i];
    double qn[4];
    quat_box_plus(pose + 3, dp + 3, qn);
    for (int i = 0; i < 4; i++) pose[3 + i] = qn[i];
    const double nrm = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2] + dp[3] * dp[3] + dp[4] * dp[4] + dp[5] * dp[5]);
    if (nrm < P.tol) break; // :72-76
  }
  if (iters_out) *iters_out = k;
  return status;
}


// ---- PoseOptimizationQP::optimize (PoseOptimizationQP.cpp:42-140): position only ------------------
typedef GiLayout<3, 1, 4> PoseQpGi;

template <class Scr>
QL_HD int pose_qp(const PoseParamsDev &P, const PoseProblem &pb, Scr &s, double pose[7]) {
  typedef PoseQpGi Ly;
  double R[9], q[3] = {0, 0, 0};
  quat_to_matrix(pose + 3, R);
  int nl = 0;
  QL_UNROLL for (int k = 0; k < 4; k++) {
    if (!((pb.present >> k) & 1u)) continue;
    nl++;
    double Rd[3];
    rot(R, pb.nominal[k], Rd);
    for (int i = 0; i < 3; i++) q[i] += -2.0 * (pb.stance[k][i] - Rd[i]);
  }
  for (int i = 0; i < 9; i++) s.at(Ly::G + i) = 0.0;
  for (int i = 0; i < 3; i++) { s.at(Ly::G + 4 * i) = 2.0 * nl; s.at(Ly::G0 + i) = q[i]; s.at(Ly::CE + i) = 0.0; }
  s.at(Ly::CE0) = 0.0;
  double GA[4][2], gb[4], Rr[3];
  const int m = polygon_halfspaces(pb.n_vertices, pb.polygon, GA, gb);
  rot(R, pb.r_com, Rr);
  QL_UNROLL for (int i = 0; i < 4; i++) {
    if (i >= m) continue;
    s.at(Ly::CI0 + i) = gb[i] - (GA[i][0] * Rr[0] + GA[i][1] * Rr[1]);
    s.at(Ly::CI + 0 * m + i) = -GA[i][0];
    s.at(Ly::CI + 1 * m + i) = -GA[i][1];
    s.at(Ly::CI + 2 * m + i) = -0.0;
  }
  double f;
  const int st = gi_solve<3, 1, 4>(s, 3, P.dummy_equality ? 1 : 0, m, &f, nullptr);
  if (st == kStatusOk)
    for (int i = 0; i < 3; i++) pose[i] = s.at(Ly::X + i);
  return st;
}

// Register-resident variant (gi6_core.hpp): the one the kernel runs.
template <class Scr>
QL_HD int pose_sqp6(const PoseParamsDev &P, const PoseProblem &pb, Scr &s, double pose[7], int *iters_out) {
  double centroid[2], GA[4][2], gb[4];
  polygon_centroid(pb.n_vertices, pb.polygon, centroid);
  const int nsp = polygon_halfspaces(pb.n_vertices, pb.polygon, GA, gb);
  int k = 0, status = kStatusOk;
  while (k < P.max_iter) {
    double G[36], g0[6], dp[6], f;
    const double CE[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int m = pose_linearise_to<Gi6Layout::CI, Gi6Layout::CI0>(
        P, pb, centroid, nsp, GA, gb, pose, s, [&](int i, double v) { G[i] = v; }, [&](int i, double v) { g0[i] = v; });
    k++;
    status = gi6_solve(s, G, g0, CE, 0.0, P.dummy_equality ? 1 : 0, m, dp, &f);
    if (status != kStatusOk) break;
    for (int i = 0; i < 3; i++) pose[i] += dp[i];
    double qn[4];
    quat_box_plus(pose + 3, dp + 3, qn);
    for (int i = 0; i < 4; i++) pose[3 + i] = qn[i];
    const double nrm = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2] + dp[3] * dp[3] + dp[4] * dp[4] + dp[5] * dp[5]);
    if (nrm < P.tol) break; // :72-76
  }
  if (iters_out) *iters_out = k;
  return status;
}

// ---- BaseAuto::optimizePose (BaseAuto.cpp:394-400): geometric -> QP -> check -> SQP -----------------
// stage: 2 = the QP result passed the checker, 3 = the SQP ran.  The scratch must hold
// max(PoseQpGi::kTotal, Gi6Layout::kTotal) doubles.
template <class Scr>
QL_HD int base_auto_optimize_pose(const PoseParamsDev &P, const PoseProblem &pb, const double sfo[4][3],
                                  const double min_len[4], double leg_tol, Scr &s, double pose[7], int *stage,
                                  int *iters_out) {
  pose_geometric(pb, sfo, pose);
  *stage = 2;
  if (iters_out) *iters_out = 0;
  int st = pose_qp(P, pb, s, pose);
  if (st != kStatusOk) return st;
  if (pose_check(pb, pose, min_len, leg_tol)) return kStatusOk;
  *stage = 3;
  return pose_sqp6(P, pb, s, pose, iters_out);
}

} // namespace qlamd
