// Per-problem arithmetic of the batched pose optimisation (BASELINE config 5), host/device.
//
// Replaces free_gait::PoseOptimizationSQP::optimize
// (free_gait_core/src/pose_optimization/PoseOptimizationSQP.cpp:58-111):
//   objective gradient / Hessian  PoseOptimizationObjectiveFunction.cpp:150-248
//   constraint values / Jacobian  PoseOptimizationFunctionConstraints.cpp:95-194
//   SQP loop                      qp_solver/src/sequencequadraticproblemsolver.cpp:18-102
//   params (+) dp                 poseparameterization.cpp:37-51
// Inner QP: qp6_coop in pose_coop.hpp (n = 6, m <= 8, one all-zero equality column as the reference passes).  The
// one-lane-per-problem drivers around the step-by-step restatement of solve_quadprog live with the host mirror
// (tests/host_mirror/pose_one_lane.hpp): they are test infrastructure, not part of the library.
#pragma once

#include <math.h>
#include <stdint.h>

#include "balance_core.hpp" // QL_HD, status codes, small vector helpers

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

QL_HD void pose_geometric_finish(const PoseProblem &pb, const double sfo[4][3], const double cen[2], double z,
                                 const double q[4], double pose[7]);

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
  pose_geometric_finish(pb, sfo, cen, z, q, pose);
}

// heading along the fore / hind mid-point line and 70 % of the roll / pitch of q (PoseOptimizationGeometric.cpp:76-97)
QL_HD void pose_geometric_finish(const PoseProblem &pb, const double sfo[4][3], const double cen[2], double z,
                                 const double q[4], double pose[7]) {
  (void)pb;
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

} // namespace qlamd
