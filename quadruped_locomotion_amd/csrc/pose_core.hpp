// Per-problem arithmetic of the batched pose optimisation (BASELINE config 5), host/device.
//
// Replaces free_gait::PoseOptimizationSQP::optimize
// (free_gait_core/src/pose_optimization/PoseOptimizationSQP.cpp:58-111):
//   objective gradient / Hessian  PoseOptimizationObjectiveFunction.cpp:150-248
//   constraint values / Jacobian  PoseOptimizationFunctionConstraints.cpp:95-194
//   SQP loop                      qp_solver/src/sequencequadraticproblemsolver.cpp:18-102
//   params (+) dp                 poseparameterization.cpp:37-51
// Inner QP: gi_core.hpp (n = 6, m <= 8, one all-zero equality column as the reference passes).
#pragma once

#include "gi6_core.hpp"
#include "gi_core.hpp"

namespace qlamd {

struct PoseParamsDev {
  double hips[4][3];   // base -> hip in base (adapter.getPositionBaseToHipInBaseFrame)
  double com_weight;   // 2.0, PoseOptimizationObjectiveFunction.cpp:17
  double tol;          // 0.05, PoseOptimizationSQP.cpp:99
  int max_iter;        // 30
  int dummy_equality;  // 1 = reference behaviour
  int leg_order[4];    // iteration order of the reference's unordered_map Stance
};

// One problem.  The per-leg arrays are stored in ITERATION order (slot k holds limb leg_order[k]),
// permuted once at load time, so that every later index is a compile-time constant (a run-time
// limb index into a register array would send the whole struct to scratch memory).
struct PoseProblem {
  double stance[4][3], nominal[4][3], hips[4][3], max_len[4], polygon[4][2], r_com[3], pose[7];
  int n_vertices;
  unsigned present; // bit k: slot k is part of the stance
};

// Fill slot k from limb-indexed batch arrays (get(l, a) style accessors keep this usable on host and device).
template <class FS, class FN, class F1, class FM>
QL_HD void pose_problem_load_legs(const PoseParamsDev &P, PoseProblem &pb, FS stance, FN nominal, F1 maxlen, FM limb_mask_of) {
  // gather first, store second: when pb lives in LDS (device) the loads of all four legs are then in flight
  // together instead of load - wait - store, leg after leg
  double st[4][3], nm[4][3], ml[4];
  QL_UNROLL for (int k = 0; k < 4; k++) {
    const int l = P.leg_order[k];
    QL_UNROLL for (int a = 0; a < 3; a++) { st[k][a] = stance(l, a); nm[k][a] = nominal(l, a); }
    ml[k] = maxlen(l);
  }
  const unsigned limb_mask = limb_mask_of(); // evaluated after the gather: may depend on a load of its own
  unsigned present = 0;
  QL_UNROLL for (int k = 0; k < 4; k++) {
    const int l = P.leg_order[k];
    if ((limb_mask >> l) & 1u) present |= 1u << k;
    QL_UNROLL for (int a = 0; a < 3; a++) {
      pb.stance[k][a] = st[k][a];
      pb.nominal[k][a] = nm[k][a];
      pb.hips[k][a] = P.hips[l][a];
    }
    pb.max_len[k] = ml[k];
  }
  pb.present = present;
}

typedef GiLayout<6, 1, 8> PoseGi;

QL_HD void skew3(const double r[3], double S[9]) {
  S[0] = 0;     S[1] = -r[2]; S[2] = r[1];
  S[3] = r[2];  S[4] = 0;     S[5] = -r[0];
  S[6] = -r[1]; S[7] = r[0];  S[8] = 0;
}
QL_HD void mm3(const double A[9], const double B[9], double C[9]) {
  QL_UNROLL for (int i = 0; i < 3; i++)
    QL_UNROLL for (int j = 0; j < 3; j++)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// The polygon helpers index the vertex array with compile-time constants only (loops unrolled to the maximum of
// four vertices and predicated on nv): a run-time index would send the whole problem struct to scratch memory.
// grid_map::Polygon::getCentroid (grid_map_core, restated; vertices counter-clockwise)
QL_HD void polygon_centroid(int nv, const double poly[4][2], double c[2]) {
  double area = 0.0;
  c[0] = c[1] = 0.0;
  QL_UNROLL for (int i = 0; i < 4; i++) {
    if (i >= nv) continue;
    double nx[2]; // vertex (i + 1) mod nv
    if (i + 1 < 4 && i + 1 != nv) { nx[0] = poly[i + 1 < 4 ? i + 1 : 0][0]; nx[1] = poly[i + 1 < 4 ? i + 1 : 0][1]; }
    else { nx[0] = poly[0][0]; nx[1] = poly[0][1]; }
    const double cr = poly[i][0] * nx[1] - nx[0] * poly[i][1];
    area += cr;
    c[0] += cr * (poly[i][0] + nx[0]);
    c[1] += cr * (poly[i][1] + nx[1]);
  }
  area *= 0.5;
  c[0] /= (6.0 * area);
  c[1] /= (6.0 * area);
}

// grid_map::Polygon::convertToInequalityConstraints: A x <= b, one row per non-degenerate edge
QL_HD int polygon_halfspaces(int nv, const double poly[4][2], double A[4][2], double b[4]) {
  double c[2] = {0, 0};
  QL_UNROLL for (int i = 0; i < 4; i++)
    if (i < nv) { c[0] += poly[i][0]; c[1] += poly[i][1]; }
  c[0] /= nv; c[1] /= nv;
  int rows = 0;
  QL_UNROLL for (int i = 0; i < 4; i++) {
    if (i >= nv) continue;
    double nx[2];
    if (i + 1 < 4 && i + 1 != nv) { nx[0] = poly[i + 1 < 4 ? i + 1 : 0][0]; nx[1] = poly[i + 1 < 4 ? i + 1 : 0][1]; }
    else { nx[0] = poly[0][0]; nx[1] = poly[0][1]; }
    const double x1 = poly[i][0] - c[0], y1 = poly[i][1] - c[1];
    const double x2 = nx[0] - c[0], y2 = nx[1] - c[1];
    const double det = x1 * y2 - x2 * y1;
    if (fabs(det) <= 1e-12 * (fabs(x1 * y2) + fabs(x2 * y1) + 1e-300)) continue;
    const double a0 = (y2 - y1) / det, a1 = (x1 - x2) / det;
    const double bb = 1.0 + (a0 * c[0] + a1 * c[1]);
    QL_UNROLL for (int r = 0; r < 4; r++)
      if (r == rows) { A[r][0] = a0; A[r][1] = a1; b[r] = bb; }
    rows++;
  }
  return rows;
}

// q.boxPlus(d) = exp(d) * q   (kindr exp map)
QL_HD void quat_box_plus(const double q[4], const double d[3], double out[4]) {
  const double v = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double e[4];
  if (v < 1e-12) {
    e[0] = 1.0; e[1] = 0.5 * d[0]; e[2] = 0.5 * d[1]; e[3] = 0.5 * d[2];
  } else {
    const double a = 0.5 * v, sn = sin(a) / v;
    e[0] = cos(a); e[1] = sn * d[0]; e[2] = sn * d[1]; e[3] = sn * d[2];
  }
  out[0] = e[0] * q[0] - e[1] * q[1] - e[2] * q[2] - e[3] * q[3];
  out[1] = e[0] * q[1] + e[1] * q[0] + e[2] * q[3] - e[3] * q[2];
  out[2] = e[0] * q[2] - e[1] * q[3] + e[2] * q[0] + e[3] * q[1];
  out[3] = e[0] * q[3] + e[1] * q[2] - e[2] * q[1] + e[3] * q[0];
}

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

// ---- PoseConstraintsChecker::check (PoseConstraintsChecker.cpp:29-64) --------------------------------
QL_HD bool polygon_is_inside(int nv, const double poly[4][2], const double pt[2]) { // grid_map crossing number
  // edges (v_j, v_i) with j = i - 1, and (v_{nv-1}, v_0) for i = 0; the latter is picked up at the unrolled
  // step idx == nv - 1 so that every vertex index is a constant (crossings are only counted: order is free)
  int cross = 0;
  const auto edge = [&](double xj, double yj, double xi, double yi) {
    if (((yi > pt[1]) != (yj > pt[1])) && (pt[0] < (xj - xi) * (pt[1] - yi) / (yj - yi) + xi)) cross++;
  };
  QL_UNROLL for (int i = 0; i < 4; i++) {
    if (i >= nv) continue;
    if (i > 0) edge(poly[i > 0 ? i - 1 : 0][0], poly[i > 0 ? i - 1 : 0][1], poly[i][0], poly[i][1]);
    if (i == nv - 1) edge(poly[i][0], poly[i][1], poly[0][0], poly[0][1]);
  }
  return (cross & 1) != 0;
}

QL_HD bool pose_check(const PoseProblem &pb, const double pose[7], const double min_len[4], double leg_tol) {
  double R[9], Pr[3];
  quat_to_matrix(pose + 3, R);
  rot(R, pb.r_com, Pr);
  const double com[2] = {pose[0] + Pr[0], pose[1] + Pr[1]};
  bool ok = polygon_is_inside(pb.n_vertices, pb.polygon, com);
  QL_UNROLL for (int k = 0; k < 4; k++) {
    if (!((pb.present >> k) & 1u)) continue;
    const double df[3] = {pb.stance[k][0] - pose[0], pb.stance[k][1] - pose[1], pb.stance[k][2] - pose[2]};
    double bf[3];
    irot(R, df, bf);
    const double e[3] = {bf[0] - pb.hips[k][0], bf[1] - pb.hips[k][1], bf[2] - pb.hips[k][2]};
    const double len = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    if (len < min_len[k] - leg_tol || len > pb.max_len[k] + leg_tol) ok = false;
  }
  return ok;
}

// ---- PoseOptimizationGeometric::optimize (PoseOptimizationGeometric.cpp:34-105) -----------------------
// Cyclic Jacobi on a symmetric 4x4 (the reference calls Eigen::EigenSolver, :69); fully unrolled so that
// A and V stay in registers.
QL_HD void sym4_eigen(const double C[16], double w[4], double V[16]) {
  double A[16];
  QL_UNROLL for (int i = 0; i < 4; i++)
    QL_UNROLL for (int j = 0; j < 4; j++) {
      A[4 * i + j] = 0.5 * (C[4 * i + j] + C[4 * j + i]);
      V[4 * i + j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 12; sweep++) {
    // converged when the off-diagonal mass is below 1e-20 of the diagonal's (quadratic convergence: 4-6 sweeps)
    double off = 0.0, dia = 0.0;
    QL_UNROLL for (int p = 0; p < 4; p++) {
      dia += fabs(A[5 * p]);
      QL_UNROLL for (int q = p + 1; q < 4; q++) off += fabs(A[4 * p + q]);
    }
    if (off <= 1e-20 * dia) break;
    QL_UNROLL for (int p = 0; p < 3; p++)
      QL_UNROLL for (int q = p + 1; q < 4; q++) {
        const double apq = A[4 * p + q];
        if (apq == 0.0) continue;
        const double theta = (A[4 * q + q] - A[4 * p + p]) * (0.5 * ql_rcp(apq));
        const double th1 = theta * theta + 1.0;
        const double t = (theta >= 0.0 ? 1.0 : -1.0) * ql_rcp(fabs(theta) + th1 * ql_rsqrt(th1));
        const double c = ql_rsqrt(t * t + 1.0), sn = t * c;
        QL_UNROLL for (int k = 0; k < 4; k++) {
          const double akp = A[4 * k + p], akq = A[4 * k + q];
          A[4 * k + p] = c * akp - sn * akq;
          A[4 * k + q] = sn * akp + c * akq;
          const double vkp = V[4 * k + p], vkq = V[4 * k + q];
          V[4 * k + p] = c * vkp - sn * vkq;
          V[4 * k + q] = sn * vkp + c * vkq;
        }
        QL_UNROLL for (int k = 0; k < 4; k++) {
          const double apk = A[4 * p + k], aqk = A[4 * q + k];
          A[4 * p + k] = c * apk - sn * aqk;
          A[4 * q + k] = sn * apk + c * aqk;
        }
      }
  }
  QL_UNROLL for (int i = 0; i < 4; i++) w[i] = A[5 * i];
}

QL_HD void quat_mul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

QL_HD void quat_log(const double d[4], double out[3]) { // rotation vector of d (boxMinus identity)
  const double s2 = 1.0 - d[0] * d[0];
  double k = 2.0;
  if (s2 >= 1e-12) k = 2.0 * acos(d[0]) / sqrt(s2);
  out[0] = k * d[1]; out[1] = k * d[2]; out[2] = k * d[3];
}

// sfo: stance for orientation by limb id (LF, RF, RH, LH).  pb holds the legs in iteration order.
QL_HD void pose_geometric(const PoseProblem &pb, const double sfo[4][3], double pose[7]) {
  double cen[2];
  polygon_centroid(pb.n_vertices, pb.polygon, cen);
  double z = 0.0, Cm[16], Am[16];
  QL_UNROLL for (int i = 0; i < 16; i++) { Cm[i] = 0.0; Am[i] = 0.0; }
  int nl = 0;
  QL_UNROLL for (int k = 0; k < 4; k++) {
    if (!((pb.present >> k) & 1u)) continue;
    nl++;
    z += pb.stance[k][2] - pb.nominal[k][2];
    const double *a = pb.stance[k], *b = pb.nominal[k];
    // Ak = Q(0,a) - Qc(0,b): left minus right quaternion-product matrix (:63-65)
    const double Ak[16] = {0.0,         -a[0] + b[0], -a[1] + b[1], -a[2] + b[2],
                           a[0] - b[0], 0.0,          -a[2] - b[2], a[1] + b[1],
                           a[1] - b[1], a[2] + b[2],  0.0,          -a[0] - b[0],
                           a[2] - b[2], -a[1] - b[1], a[0] + b[0],  0.0};
    QL_UNROLL for (int i = 0; i < 4; i++)
      QL_UNROLL for (int j = 0; j < 4; j++) {
        double acc = 0.0;
        QL_UNROLL for (int m = 0; m < 4; m++) acc += Ak[4 * i + m] * Ak[4 * m + j];
        Cm[4 * i + j] += acc;
        Am[4 * i + j] += Ak[4 * i + j];
      }
  }
  const double n = (double)nl, rn = ql_rcp(n);
  z *= rn;
  QL_UNROLL for (int i = 0; i < 16; i++) Am[i] = Am[i] * rn;
  QL_UNROLL for (int i = 0; i < 4; i++)
    QL_UNROLL for (int j = 0; j < 4; j++) {
      double acc = 0.0;
      QL_UNROLL for (int m = 0; m < 4; m++) acc += Am[4 * i + m] * Am[4 * m + j];
      Cm[4 * i + j] -= n * acc;
    }
  double w[4], V[16];
  sym4_eigen(Cm, w, V);
  double wb = w[0], q[4] = {V[0], V[4], V[8], V[12]};
  QL_UNROLL for (int i = 1; i < 4; i++)
    if (w[i] > wb) { wb = w[i]; q[0] = V[i]; q[1] = V[4 + i]; q[2] = V[8 + i]; q[3] = V[12 + i]; }
  const double inq = ql_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  QL_UNROLL for (int i = 0; i < 4; i++) q[i] *= inq;
  { // setUnique: first non-zero component positive
    bool neg = false, decided = false;
    QL_UNROLL for (int i = 0; i < 4; i++)
      if (!decided && q[i] != 0.0) { decided = true; neg = q[i] < 0.0; }
    if (neg) {
      QL_UNROLL for (int i = 0; i < 4; i++) q[i] = -q[i];
    }
  }
  // heading (:76-81), Eigen setFromTwoVectors(UnitX, dir)
  double dir[3];
  QL_UNROLL for (int i = 0; i < 3; i++) dir[i] = 0.5 * (sfo[0][i] + sfo[1][i]) - 0.5 * (sfo[3][i] + sfo[2][i]);
  dir[2] = 0.0;
  double heading[4];
  {
    const double inb = ql_rsqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    const double v1[3] = {dir[0] * inb, dir[1] * inb, dir[2] * inb};
    double c = v1[0]; // v0 = (1, 0, 0) after normalisation
    if (c < -1.0 + 1e-12) {
      c = c > -1.0 ? c : -1.0;
      const double w2 = (1.0 + c) * 0.5;
      heading[0] = sqrt(w2); heading[1] = 0.0; heading[2] = 0.0; heading[3] = sqrt(1.0 - w2);
    } else {
      const double invs = ql_rsqrt((1.0 + c) * 2.0), sc = (1.0 + c) * 2.0 * invs;
      heading[0] = sc * 0.5;
      heading[1] = (0.0 * v1[2] - 0.0 * v1[1]) * invs;
      heading[2] = (0.0 * v1[0] - 1.0 * v1[2]) * invs;
      heading[3] = (1.0 * v1[1] - 0.0 * v1[0]) * invs;
    }
  }
  // 70 % of roll/pitch (:84-87)
  const double ident[4] = {1.0, 0.0, 0.0, 0.0};
  double rv[3], yaw[4], rel[4], rp[4];
  quat_log(q, rv);
  const double rvz[3] = {0.0, 0.0, rv[2]};
  quat_box_plus(ident, rvz, yaw);
  const double yaw_inv[4] = {yaw[0], -yaw[1], -yaw[2], -yaw[3]};
  quat_mul(yaw_inv, q, rel);
  quat_log(rel, rv);
  QL_UNROLL for (int i = 0; i < 3; i++) rv[i] *= 0.7;
  quat_box_plus(ident, rv, rp);
  pose[0] = cen[0]; pose[1] = cen[1]; pose[2] = z;
  quat_mul(heading, rp, pose + 3);
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
