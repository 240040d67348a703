// Per-robot arithmetic of the batched balance-controller step.
//
// One call = one control tick of one robot:
//   virtual-model wrench  -> VirtualModelController.cpp:104-268
//   leg FK                -> quadrupedkinematics.cpp:143-212
//   QP assembly           -> ContactForceDistribution.cpp:138-336
//   QP solve              -> ContactForceDistribution.cpp:385-514 (ooqpei there)
//   joint torques + clamp -> ContactForceDistribution.cpp:516-578,
//                            ros_balance_controller.cpp:451-454
//
// The code is written for one robot per lane: everything a lane indexes with a
// compile-time constant lives in VGPRs (the 12x12 Cholesky factor of the QP
// Hessian included); the few arrays that need run-time indices (the Cholesky
// factor of the active-set Schur complement, the multipliers) go through a
// `Scratch` object that the kernel maps onto LDS with a [element][lane] layout
// (bank = lane, so a per-lane index never conflicts).
//
// QP method.  The reference solves the force-distribution QP with OOQP; its
// own dense solver is Goldfarb-Idnani (QuadProg++).  We keep the G-I iteration
// (most-violated constraint, partial/full steps, drops) but replace its dense
// J/R updates by the range-space form: with G = L L' fixed,
//   g = G^-1 n_p,  r = (N'G^-1 N)^-1 N' g,  z = G^-1 (n_p - N r),
// and the Cholesky factor C of the q x q Schur complement N'G^-1 N is grown /
// shrunk as constraints enter / leave.  Every constraint normal touches one
// leg only (3 non-zeros), which makes N' g and N r a handful of flops.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define QL_HD __host__ __device__ __forceinline__
#define QL_UNROLL _Pragma("unroll")
#else
#define QL_HD inline
#define QL_UNROLL
#endif

namespace qlamd {

constexpr int kStatusOk = 0, kStatusInfeasible = 1, kStatusNotPd = 2, kStatusMaxIter = 3;
constexpr int kMaxOuter = 200; // guard; the reference has none

// Batch-invariant data, resident in device memory, read through scalar loads.
struct DeviceParams {
  double kp_t[3], kd_t[3], kff_t[3];
  double kp_r[3], kd_r[3], kff_r[3];
  double S[6];
  double w_reg, mu, f_min, tau_max;
  double grav;                 // g_W = (0,0,-grav)
  double Fg_scale;             // grav_comp * (torso + sum legs)  : F_g = -Fg_scale * g_B
  double Tg_arm[3];            // sum_l grav_comp*m_l*(hip_l - com) + grav_comp*m_torso*com : T_g = -(Tg_arm x g_B)
  // leg chains: per [leg][segment] fixed rotation R0 (row-major), origin, mass, mass*com
  double R0[4][4][9];
  double xyz[4][4][3];
  double mass[4][4];
  double mcom[4][4][3];        // mass * com (link frame)
  int rot_is_identity[4][4];   // R0 == I exactly (skips a 3x3 product)
};

struct RobotIn {
  double q[12];
  double pos[3], quat[4], linvel[3], angvel[3];
  double dpos[3], dquat[4], dlinvel[3], dangvel[3];
  unsigned stance; // bit l = leg l supports
};

// ------------------------------------------------------------ small math ---

QL_HD void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
QL_HD double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// sin and cos for |x| up to a few thousand (joint angles are within +-3 rad,
// quadruped_model.urdf:53-57): Cody-Waite reduction by pi/2 in two parts, then
// the classical minimax kernels on [-pi/4, pi/4] (error < 1 ulp).
QL_HD void sincos_pio4(double x, double &s, double &c) {
  const double z = x * x;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
  s = x + x * z * ps;
  const double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  c = 1.0 - (0.5 * z - z * pc);
}
QL_HD void sincos_reduced(double x, double &s, double &c) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00; // first 33 bits of pi/2
  const double pio2_lo = 6.07710050650619224932e-11; // pi/2 - pio2_hi
  const double kf = rint(x * two_over_pi);
  const int k = (int)kf;
  const double r = (x - kf * pio2_hi) - kf * pio2_lo;
  double sr, cr;
  sincos_pio4(r, sr, cr);
  const bool swap = (k & 1) != 0;
  const double s0 = swap ? cr : sr, c0 = swap ? sr : cr;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}

// RotationQuaternion(w,x,y,z) -> matrix, Eigen's toRotationMatrix form (kindr).
QL_HD void quat_to_matrix(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
QL_HD void rot(const double R[9], const double v[3], double o[3]) {
  QL_UNROLL for (int i = 0; i < 3; i++) o[i] = R[i * 3] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2];
}
QL_HD void irot(const double R[9], const double v[3], double o[3]) {
  QL_UNROLL for (int i = 0; i < 3; i++) o[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}

// -(q_d^-1).boxMinus(q_m^-1) = -log(q_d^-1 * q_m)  (VirtualModelController.cpp:120-124)
QL_HD void orientation_error(const double qd[4], const double qm[4], double e[3]) {
  // d = conj(qd) * qm
  const double aw = qd[0], ax = -qd[1], ay = -qd[2], az = -qd[3];
  const double dw = aw * qm[0] - ax * qm[1] - ay * qm[2] - az * qm[3];
  const double dx = aw * qm[1] + ax * qm[0] + ay * qm[3] - az * qm[2];
  const double dy = aw * qm[2] - ax * qm[3] + ay * qm[0] + az * qm[1];
  const double dz = aw * qm[3] + ax * qm[2] - ay * qm[1] + az * qm[0];
  const double s2 = 1.0 - dw * dw;
  double k = 2.0;
  if (s2 >= 1e-12) k = 2.0 * acos(dw) / sqrt(s2);
  e[0] = -k * dx; e[1] = -k * dy; e[2] = -k * dz;
}

// ------------------------------------------------------- virtual wrench ----

QL_HD void virtual_wrench(const DeviceParams &P, const RobotIn &in, const double Rm[9], const double gB[3],
                          double b[6]) {
  double e_p[3], e_v[3], e_w[3], e_o[3];
  QL_UNROLL for (int i = 0; i < 3; i++) {
    e_p[i] = in.dpos[i] - in.pos[i];
    e_v[i] = in.dlinvel[i] - in.linvel[i];
    e_w[i] = in.dangvel[i] - in.angvel[i];
  }
  orientation_error(in.dquat, in.quat, e_o);
  // VirtualModelController.cpp:208-231 (vertical P and D enter twice)
  const double ff[3] = {in.dlinvel[0], in.dlinvel[1], 0.0};
  const double gfb[3] = {0.0, 0.0, P.kp_t[2] * e_p[2]};
  const double gdb[3] = {0.0, 0.0, P.kd_t[2] * e_v[2]};
  double Rep[3], Rev[3], Rff[3], fbp[3], fbd[3];
  irot(Rm, e_p, Rep); irot(Rm, e_v, Rev); irot(Rm, ff, Rff); irot(Rm, gfb, fbp); irot(Rm, gdb, fbd);
  QL_UNROLL for (int i = 0; i < 3; i++)
    b[i] = P.kp_t[i] * Rep[i] + P.kd_t[i] * Rev[i] + P.kff_t[i] * Rff[i] - P.Fg_scale * gB[i] + fbp[i] + fbd[i];
  // VirtualModelController.cpp:244-259
  const double kdw[3] = {P.kd_r[0] * e_w[0], P.kd_r[1] * e_w[1], P.kd_r[2] * e_w[2]};
  const double kfw[3] = {0.0, 0.0, P.kff_r[2] * in.dangvel[2]};
  double Rd[3], Rf[3], Tg[3];
  irot(Rm, kdw, Rd); irot(Rm, kfw, Rf);
  cross3(P.Tg_arm, gB, Tg);
  QL_UNROLL for (int i = 0; i < 3; i++) b[3 + i] = P.kp_r[i] * e_o[i] + Rd[i] + Rf[i] - Tg[i];
}

// ------------------------------------------------------- leg kinematics ----

// Cumulative frames of one leg chain.  Rc[k], pc[k]: pose of link k in base.
struct LegFrames {
  double R[4][9];
  double p[4][3];
};

QL_HD void leg_frames(const DeviceParams &P, int leg, const double q[3], LegFrames &F) {
  double Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double pc[3] = {0, 0, 0};
  QL_UNROLL for (int k = 0; k < 4; k++) {
    const double *R0 = P.R0[leg][k];
    const double *t = P.xyz[leg][k];
    double Rs[9]; // segment rotation R0 * Rz(q)
    if (k < 3) {
      double s, c;
      sincos_reduced(q[k], s, c);
      QL_UNROLL for (int i = 0; i < 3; i++) {
        Rs[i * 3 + 0] = R0[i * 3 + 0] * c + R0[i * 3 + 1] * s;
        Rs[i * 3 + 1] = R0[i * 3 + 1] * c - R0[i * 3 + 0] * s;
        Rs[i * 3 + 2] = R0[i * 3 + 2];
      }
    } else {
      QL_UNROLL for (int i = 0; i < 9; i++) Rs[i] = R0[i];
    }
    double pn[3], Rn[9];
    QL_UNROLL for (int i = 0; i < 3; i++)
      pn[i] = pc[i] + (Rc[i * 3] * t[0] + Rc[i * 3 + 1] * t[1] + Rc[i * 3 + 2] * t[2]);
    QL_UNROLL for (int i = 0; i < 3; i++)
      QL_UNROLL for (int j = 0; j < 3; j++)
        Rn[i * 3 + j] = Rc[i * 3] * Rs[j] + Rc[i * 3 + 1] * Rs[3 + j] + Rc[i * 3 + 2] * Rs[6 + j];
    QL_UNROLL for (int i = 0; i < 9; i++) { Rc[i] = Rn[i]; F.R[k][i] = Rn[i]; }
    QL_UNROLL for (int i = 0; i < 3; i++) { pc[i] = pn[i]; F.p[k][i] = pn[i]; }
  }
}

// Translation Jacobian (column i = z_i x (p_foot - p_i)) and gravity torque
// G_i = -g . (z_i x sum_{k>=i} m_k (c_k - p_i))  from the frames.
QL_HD void leg_jac_grav(const DeviceParams &P, int leg, const LegFrames &F, const double g[3],
                        double J[9], double Gq[3]) {
  // mass-weighted COM offsets of the links in base coordinates
  double mc[4][3];
  QL_UNROLL for (int k = 0; k < 4; k++) {
    const double *c = P.mcom[leg][k];
    const double m = P.mass[leg][k];
    QL_UNROLL for (int i = 0; i < 3; i++)
      mc[k][i] = m * F.p[k][i] + (F.R[k][i * 3] * c[0] + F.R[k][i * 3 + 1] * c[1] + F.R[k][i * 3 + 2] * c[2]);
  }
  // suffix sums over distal links: M_i = sum_{k>=i} m_k, H_i = sum_{k>=i} m_k c_k
  double H[3] = {mc[3][0], mc[3][1], mc[3][2]};
  double M = P.mass[leg][3];
  QL_UNROLL for (int i = 2; i >= 0; i--) {
    QL_UNROLL for (int a = 0; a < 3; a++) H[a] += mc[i][a];
    M += P.mass[leg][i];
    const double z[3] = {F.R[i][2], F.R[i][5], F.R[i][8]};
    const double d[3] = {F.p[3][0] - F.p[i][0], F.p[3][1] - F.p[i][1], F.p[3][2] - F.p[i][2]};
    double v[3];
    cross3(z, d, v);
    J[0 * 3 + i] = v[0]; J[1 * 3 + i] = v[1]; J[2 * 3 + i] = v[2];
    const double h[3] = {H[0] - M * F.p[i][0], H[1] - M * F.p[i][1], H[2] - M * F.p[i][2]};
    double zh[3];
    cross3(z, h, zh);
    Gq[i] = -dot3(g, zh);
  }
}

// -------------------------------------------------------------- the QP -----

// Packed lower-triangular index.
QL_HD constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// Constraint ids: id = 5*leg + t;  t = 0: n.f >= f_min;  t = 1..4: (mu n +- t1/t2).f >= 0
// in the reference's row order +t1, -t1, +t2, -t2 (ContactForceDistribution.cpp:314-325).
QL_HD int id_leg(int id) { return (id * 13) >> 6; } // id / 5 for id < 20

struct Pyramid { // friction pyramid directions in the base frame (one set; see kernel for per-leg normals)
  double n[3], t1[3], t2[3];
};

QL_HD void constraint_normal(const Pyramid &py, double mu, int t, double nv[3]) {
  const bool first = (t == 1) || (t == 2);
  const double sgn = (t == 1 || t == 3) ? 1.0 : -1.0;
  QL_UNROLL for (int a = 0; a < 3; a++) {
    const double tang = first ? py.t1[a] : py.t2[a];
    nv[a] = (t == 0) ? py.n[a] : (mu * py.n[a] + sgn * tang);
  }
}

// y = L^-1 b, then x = L^-T y.  L packed lower, inv_d[i] = 1 / L_ii.
QL_HD void chol_solve12(const double L[78], const double inv_d[12], const double b[12], double x[12]) {
  double y[12];
  QL_UNROLL for (int i = 0; i < 12; i++) {
    double acc = b[i];
    QL_UNROLL for (int j = 0; j < i; j++) acc -= L[tri(i, j)] * y[j];
    y[i] = acc * inv_d[i];
  }
  QL_UNROLL for (int i = 11; i >= 0; i--) {
    double acc = y[i];
    QL_UNROLL for (int j = i + 1; j < 12; j++) acc -= L[tri(j, i)] * x[j];
    x[i] = acc * inv_d[i];
  }
}

// Scratch: run-time indexed per-robot arrays.  Layout of the element space:
//   [0,78)    C   packed lower Cholesky factor of the Schur complement
//   [78,91)   u   multipliers of the active constraints (+1 slot for the candidate)
//   [91,104)  r   dual step direction
//   [104,117) uo  multipliers saved at the start of an outer iteration
//   [117,129) xo  x saved at the start of an outer iteration
//   [129,141) x0  unconstrained minimiser (kept for the final polish)
constexpr int kScrC = 0, kScrU = 78, kScrR = 91, kScrUo = 104, kScrXo = 117, kScrX0 = 129, kScratchDoubles = 141;

struct HostScratch {
  double a[kScratchDoubles];
  QL_HD double &at(int e) { return a[e]; }
};

struct QpResult {
  int status;
  int iters;
  int n_active;
};

// Solve  min 1/2 x'Gx + g0'x  s.t. the pyramid / minimum-force constraints of
// the stance legs.  G (packed lower, 12x12, legs outside `stance` decoupled as
// w*I) is overwritten by its Cholesky factor.
template <bool kPerLeg>
QL_HD Pyramid pick_pyramid(const Pyramid (&py)[4], int leg) {
  if (!kPerLeg) return py[0];
  Pyramid o;
  QL_UNROLL for (int a = 0; a < 3; a++) {
    o.n[a] = leg == 0 ? py[0].n[a] : leg == 1 ? py[1].n[a] : leg == 2 ? py[2].n[a] : py[3].n[a];
    o.t1[a] = leg == 0 ? py[0].t1[a] : leg == 1 ? py[1].t1[a] : leg == 2 ? py[2].t1[a] : py[3].t1[a];
    o.t2[a] = leg == 0 ? py[0].t2[a] : leg == 1 ? py[1].t2[a] : leg == 2 ? py[2].t2[a] : py[3].t2[a];
  }
  return o;
}

template <bool kPerLeg, class Scratch>
QL_HD QpResult force_qp_solve(double G[78], const double g0[12], const Pyramid (&py)[4],
                              double mu, double f_min, unsigned stance, int nS, Scratch &scr, double x[12]) {
  QpResult res = {kStatusOk, 0, 0};
  const double eps = 2.220446049250313e-16;
  const double inf = INFINITY;

  // --- Cholesky G = L L' (in place), c1 = trace(G), c2 = trace(L^-1) over the stance block
  double c1 = 0.0, c2 = 0.0;
  double inv_d[12];
  bool not_pd = false;
  QL_UNROLL for (int i = 0; i < 12; i++)
    if ((stance >> (i / 3)) & 1u) c1 += G[tri(i, i)];
  QL_UNROLL for (int j = 0; j < 12; j++) {
    double d = G[tri(j, j)];
    QL_UNROLL for (int k = 0; k < j; k++) d -= G[tri(j, k)] * G[tri(j, k)];
    if (!(d > 0.0)) not_pd = true;
    const double ljj = sqrt(d);
    const double inv = 1.0 / ljj;
    G[tri(j, j)] = ljj;
    inv_d[j] = inv;
    if ((stance >> (j / 3)) & 1u) c2 += inv;
    QL_UNROLL for (int i = j + 1; i < 12; i++) {
      double acc = G[tri(i, j)];
      QL_UNROLL for (int k = 0; k < j; k++) acc -= G[tri(i, k)] * G[tri(j, k)];
      G[tri(i, j)] = acc * inv;
    }
  }
  if (not_pd) { res.status = kStatusNotPd; return res; }
  const double *L = G;

  // --- unconstrained minimiser x = -G^-1 g0
  {
    double ng[12];
    QL_UNROLL for (int i = 0; i < 12; i++) ng[i] = -g0[i];
    chol_solve12(L, inv_d, ng, x);
    QL_UNROLL for (int i = 0; i < 12; i++) scr.at(kScrX0 + i) = x[i];
  }

  const int m = 5 * nS;
  const double psi_tol = (double)m * eps * c1 * c2 * 100.0; // QuadProg++.cc:246
  uint64_t act = 0;      // active ids, 5 bits each, position k at bits [5k, 5k+5)
  unsigned act_mask = 0; // bit id set <=> active
  int q = 0;
  double rnorm = 1.0;

  const auto leg_pyr = [&](int leg) -> Pyramid { return pick_pyramid<kPerLeg>(py, leg); };

  // slack of constraint (leg, t) at x
  const auto slack = [&](int leg, int t, const double xx[12]) -> double {
    double xl[3];
    QL_UNROLL for (int a = 0; a < 3; a++)
      xl[a] = leg == 0 ? xx[a] : leg == 1 ? xx[3 + a] : leg == 2 ? xx[6 + a] : xx[9 + a];
    double nv[3];
    constraint_normal(leg_pyr(leg), mu, t, nv);
    return dot3(nv, xl) - (t == 0 ? f_min : 0.0);
  };

  for (int outer = 0; outer < kMaxOuter; outer++) {
    res.iters++;
    // ---- step 1: slacks of the inactive constraints, sum of infeasibilities
    double psi = 0.0, ss = 0.0;
    int ip = -1;
    unsigned excl = 0;
    // reference order: all minimum-force rows first, then the friction rows leg by leg
    QL_UNROLL for (int pass = 0; pass < 2; pass++) {
      QL_UNROLL for (int leg = 0; leg < 4; leg++) {
        if (!((stance >> leg) & 1u)) continue;
        const Pyramid &pl = py[kPerLeg ? leg : 0];
        const double dn = pl.n[0] * x[3 * leg] + pl.n[1] * x[3 * leg + 1] + pl.n[2] * x[3 * leg + 2];
        if (pass == 0) {
          const double s = dn - f_min;
          psi += fmin(0.0, s);
          if (s < ss && !((act_mask >> (5 * leg)) & 1u)) { ss = s; ip = 5 * leg; }
        } else {
          const double d1 = pl.t1[0] * x[3 * leg] + pl.t1[1] * x[3 * leg + 1] + pl.t1[2] * x[3 * leg + 2];
          const double d2 = pl.t2[0] * x[3 * leg] + pl.t2[1] * x[3 * leg + 1] + pl.t2[2] * x[3 * leg + 2];
          const double sv[4] = {mu * dn + d1, mu * dn - d1, mu * dn + d2, mu * dn - d2};
          QL_UNROLL for (int t = 1; t <= 4; t++) {
            const double s = sv[t - 1];
            psi += fmin(0.0, s);
            if (s < ss && !((act_mask >> (5 * leg + t)) & 1u)) { ss = s; ip = 5 * leg + t; }
          }
        }
      }
    }
    if (fabs(psi) <= psi_tol) break; // numerically feasible, QuadProg++.cc:246-250
    if (ip < 0) break;               // nothing violated among the inactive rows, :271-274

    // save state for the (rare) degenerate rollback, QuadProg++.cc:252-260
    const uint64_t act_old = act;
    const int q_old = q;
    for (int k = 0; k < q; k++) scr.at(kScrUo + k) = scr.at(kScrU + k);
    QL_UNROLL for (int i = 0; i < 12; i++) scr.at(kScrXo + i) = x[i];

    bool next_outer = false;
    while (!next_outer) { // ---- step 2: a candidate ip
      const int pleg = id_leg(ip), pt = ip - 5 * pleg;
      double npv[3];
      constraint_normal(leg_pyr(pleg), mu, pt, npv);
      double sp = ss;
      scr.at(kScrU + q) = 0.0;

      bool stepping = true;
      int guard = 0;
      while (stepping) { // ---- step 2a: direction; 2b: step length; 2c: step
        if (++guard > 64) { res.status = kStatusMaxIter; return res; }
        // g = G^-1 n_p
        double np12[12], g[12];
        QL_UNROLL for (int l = 0; l < 4; l++)
          QL_UNROLL for (int a = 0; a < 3; a++) np12[3 * l + a] = (l == pleg) ? npv[a] : 0.0;
        chol_solve12(L, inv_d, np12, g);
        // Schur column c = N' g; solve C y = c; r = C^-T y; v = n_p - N r
        double v[12];
        QL_UNROLL for (int i = 0; i < 12; i++) v[i] = np12[i];
        double z[12];
        if (q > 0) {
          for (int k = 0; k < q; k++) {
            const int id = (int)((act >> (5 * k)) & 31u);
            const int leg = id_leg(id), t = id - 5 * leg;
            double nk[3], gl[3];
            constraint_normal(leg_pyr(leg), mu, t, nk);
            QL_UNROLL for (int a = 0; a < 3; a++)
              gl[a] = leg == 0 ? g[a] : leg == 1 ? g[3 + a] : leg == 2 ? g[6 + a] : g[9 + a];
            double acc = dot3(nk, gl);
            for (int j = 0; j < k; j++) acc -= scr.at(kScrC + tri(k, j)) * scr.at(kScrR + j);
            scr.at(kScrR + k) = acc / scr.at(kScrC + tri(k, k)); // y_k (kept in r)
          }
          for (int k = q - 1; k >= 0; k--) {
            double acc = scr.at(kScrR + k);
            for (int j = k + 1; j < q; j++) acc -= scr.at(kScrC + tri(j, k)) * scr.at(kScrR + j);
            const double rk = acc / scr.at(kScrC + tri(k, k));
            scr.at(kScrR + k) = rk;
            const int id = (int)((act >> (5 * k)) & 31u);
            const int leg = id_leg(id), t = id - 5 * leg;
            double nk[3];
            constraint_normal(leg_pyr(leg), mu, t, nk);
            QL_UNROLL for (int l = 0; l < 4; l++) {
              const double coef = (l == leg) ? rk : 0.0;
              QL_UNROLL for (int a = 0; a < 3; a++) v[3 * l + a] -= coef * nk[a];
            }
          }
          chol_solve12(L, inv_d, v, z);
        } else {
          QL_UNROLL for (int i = 0; i < 12; i++) z[i] = g[i];
        }
        double zl[3];
        QL_UNROLL for (int a = 0; a < 3; a++)
          zl[a] = pleg == 0 ? z[a] : pleg == 1 ? z[3 + a] : pleg == 2 ? z[6 + a] : z[9 + a];
        const double zn = dot3(zl, npv);
        double zz = 0.0;
        QL_UNROLL for (int i = 0; i < 12; i++) zz += z[i] * z[i];

        // step lengths, QuadProg++.cc:304-331
        double t1 = inf;
        int lpos = -1;
        for (int k = 0; k < q; k++) {
          const double rk = scr.at(kScrR + k);
          if (rk > 0.0) {
            const double ratio = scr.at(kScrU + k) / rk;
            if (ratio < t1) { t1 = ratio; lpos = k; }
          }
        }
        double t2 = inf;
        if (fabs(zz) > eps) {
          t2 = -sp / zn;
          if (t2 < 0.0) t2 = inf;
        }
        const double t = fmin(t1, t2);
        if (t >= inf) { res.status = kStatusInfeasible; return res; } // :339-344

        const bool dual_only = (t2 >= inf);
        if (!dual_only) {
          QL_UNROLL for (int i = 0; i < 12; i++) x[i] += t * z[i];
        }
        for (int k = 0; k < q; k++) scr.at(kScrU + k) -= t * scr.at(kScrR + k);
        scr.at(kScrU + q) += t;

        if (!dual_only && fabs(t - t2) < eps) {
          // ---- full step: constraint ip becomes active, :384-421
          // new Schur row: [y', delta], y = C^-1 N'g, delta^2 = n_p'g - y'y  ( = z'n_p )
          double gl[3];
          QL_UNROLL for (int a = 0; a < 3; a++)
            gl[a] = pleg == 0 ? g[a] : pleg == 1 ? g[3 + a] : pleg == 2 ? g[6 + a] : g[9 + a];
          double dd = dot3(npv, gl);
          for (int k = 0; k < q; k++) {
            const int id = (int)((act >> (5 * k)) & 31u);
            const int leg = id_leg(id), tt = id - 5 * leg;
            double nk[3], gk[3];
            constraint_normal(leg_pyr(leg), mu, tt, nk);
            QL_UNROLL for (int a = 0; a < 3; a++)
              gk[a] = leg == 0 ? g[a] : leg == 1 ? g[3 + a] : leg == 2 ? g[6 + a] : g[9 + a];
            double acc = dot3(nk, gk);
            for (int j = 0; j < k; j++) acc -= scr.at(kScrC + tri(k, j)) * scr.at(kScrC + tri(q, j));
            const double yk = acc / scr.at(kScrC + tri(k, k));
            scr.at(kScrC + tri(q, k)) = yk;
            dd -= yk * yk;
          }
          const double delta = dd > 0.0 ? sqrt(dd) : 0.0;
          if (delta <= eps * rnorm) {
            // degenerate: exclude ip, restore the state saved at step 1, pick another, :392-412
            excl |= 1u << ip;
            act = act_old; q = q_old;
            act_mask = 0;
            for (int k = 0; k < q; k++) {
              act_mask |= 1u << (int)((act >> (5 * k)) & 31u);
              scr.at(kScrU + k) = scr.at(kScrUo + k);
            }
            QL_UNROLL for (int i = 0; i < 12; i++) x[i] = scr.at(kScrXo + i);
            // rebuild the Schur factor of the restored working set
            for (int kk = 0; kk < q; kk++) {
              const int idn = (int)((act >> (5 * kk)) & 31u);
              const int legn = id_leg(idn), tn = idn - 5 * legn;
              double nn[3], nn12[12], gn[12];
              constraint_normal(leg_pyr(legn), mu, tn, nn);
              QL_UNROLL for (int l = 0; l < 4; l++)
                QL_UNROLL for (int a = 0; a < 3; a++) nn12[3 * l + a] = (l == legn) ? nn[a] : 0.0;
              chol_solve12(L, inv_d, nn12, gn);
              double d2 = 0.0;
              for (int k = 0; k <= kk; k++) {
                const int id = (int)((act >> (5 * k)) & 31u);
                const int leg = id_leg(id), tt = id - 5 * leg;
                double nk[3], gk[3];
                constraint_normal(leg_pyr(leg), mu, tt, nk);
                QL_UNROLL for (int a = 0; a < 3; a++)
                  gk[a] = leg == 0 ? gn[a] : leg == 1 ? gn[3 + a] : leg == 2 ? gn[6 + a] : gn[9 + a];
                double acc = dot3(nk, gk);
                for (int j = 0; j < k; j++) acc -= scr.at(kScrC + tri(k, j)) * scr.at(kScrC + tri(kk, j));
                if (k < kk) {
                  const double yk = acc / scr.at(kScrC + tri(k, k));
                  scr.at(kScrC + tri(kk, k)) = yk;
                } else {
                  d2 = acc;
                }
              }
              scr.at(kScrC + tri(kk, kk)) = sqrt(fmax(d2, 0.0));
            }
            // choose the next most violated admissible row at the restored x (label l2)
            ss = 0.0; ip = -1;
            QL_UNROLL for (int pass = 0; pass < 2; pass++)
              QL_UNROLL for (int leg = 0; leg < 4; leg++) {
                if (!((stance >> leg) & 1u)) continue;
                QL_UNROLL for (int tt = (pass ? 1 : 0); tt <= (pass ? 4 : 0); tt++) {
                  const int id = 5 * leg + tt;
                  if (((act_mask | excl) >> id) & 1u) continue;
                  const double s = slack(leg, tt, x);
                  if (s < ss) { ss = s; ip = id; }
                }
              }
            if (ip < 0) return res.n_active = q, res; // :271-274
            stepping = false; // back to step 2 with the new ip
          } else {
            scr.at(kScrC + tri(q, q)) = delta;
            rnorm = fmax(rnorm, delta);
            act |= (uint64_t)ip << (5 * q);
            act_mask |= 1u << ip;
            q++;
            stepping = false;
            next_outer = true;
          }
        } else {
          // ---- partial step (or dual-only step): drop the blocking constraint, :346-362, :423-445
          // Remove row lpos of C (C C' = N'G^-1 N) and restore the triangle.  Work in the OLD
          // index space: for the column pair (c, c+1), c = lpos .. q-2, a rotation computed from
          // old row c+1 zeroes its overhang; rows below are rotated; finally rows move up by one.
          {
            for (int c = lpos; c < q - 1; c++) {
              // rotation from old row c+1: (C[c+1][c], C[c+1][c+1]) -> (h, 0)
              const double a0 = scr.at(kScrC + tri(c + 1, c));
              const double b0 = scr.at(kScrC + tri(c + 1, c + 1));
              const double h = sqrt(a0 * a0 + b0 * b0);
              const double cs = h > 0.0 ? a0 / h : 1.0, sn = h > 0.0 ? b0 / h : 0.0;
              // rows below: i = c+2 .. q-1 : (C[i][c], C[i][c+1]) -> (cs*a + sn*b, -sn*a + cs*b)
              for (int i = c + 2; i < q; i++) {
                const double a = scr.at(kScrC + tri(i, c));
                const double b = scr.at(kScrC + tri(i, c + 1));
                scr.at(kScrC + tri(i, c)) = cs * a + sn * b;
                scr.at(kScrC + tri(i, c + 1)) = -sn * a + cs * b;
              }
              scr.at(kScrC + tri(c + 1, c)) = h;
              scr.at(kScrC + tri(c + 1, c + 1)) = 0.0;
            }
            // now shift rows lpos+1 .. q-1 up by one (row i -> i-1, columns 0 .. i-1)
            for (int i = lpos + 1; i < q; i++)
              for (int j = 0; j < i; j++) scr.at(kScrC + tri(i - 1, j)) = scr.at(kScrC + tri(i, j));
          }
          // drop from the lists (u[q] is the candidate's multiplier and moves down too)
          const int id_drop = (int)((act >> (5 * lpos)) & 31u);
          act_mask &= ~(1u << id_drop);
          for (int k = lpos; k < q; k++) scr.at(kScrU + k) = scr.at(kScrU + k + 1);
          {
            const uint64_t low = act & ((1ull << (5 * lpos)) - 1ull);
            const uint64_t high = (act >> (5 * (lpos + 1))) << (5 * lpos);
            act = low | high;
          }
          q--;
          if (!dual_only) sp = slack(pleg, pt, x); // :436-440
        }
      } // step 2a loop
    }   // step 2 loop
  }     // outer
  if (res.iters >= kMaxOuter) res.status = kStatusMaxIter;
  res.n_active = q;

  // ---- final polish.  The Schur factor C carries cond(N'G^-1 N) ~ 1/w, so the iterates sit on
  // the active rows only to ~1e-7.  With the working set known, recompute x from the unconstrained
  // minimiser by the corrected semi-normal equations: u = M^-1 (b_A - N'x0), x1 = x0 + G^-1 N u,
  // then one correction with the residual of the active rows at x1.
  if (q > 0) {
    QL_UNROLL for (int i = 0; i < 12; i++) x[i] = scr.at(kScrX0 + i);
    QL_UNROLL for (int pass = 0; pass < 2; pass++) {
      double v[12], dx[12];
      QL_UNROLL for (int i = 0; i < 12; i++) v[i] = 0.0;
      // rhs_k = b_k - n_k'x ; forward solve in place (kept in r)
      for (int k = 0; k < q; k++) {
        const int id = (int)((act >> (5 * k)) & 31u);
        const int leg = id_leg(id), t = id - 5 * leg;
        double acc = -slack(leg, t, x);
        for (int j = 0; j < k; j++) acc -= scr.at(kScrC + tri(k, j)) * scr.at(kScrR + j);
        scr.at(kScrR + k) = acc / scr.at(kScrC + tri(k, k));
      }
      for (int k = q - 1; k >= 0; k--) {
        double acc = scr.at(kScrR + k);
        for (int j = k + 1; j < q; j++) acc -= scr.at(kScrC + tri(j, k)) * scr.at(kScrR + j);
        const double uk = acc / scr.at(kScrC + tri(k, k));
        scr.at(kScrR + k) = uk;
        const int id = (int)((act >> (5 * k)) & 31u);
        const int leg = id_leg(id), t = id - 5 * leg;
        double nk[3];
        constraint_normal(leg_pyr(leg), mu, t, nk);
        QL_UNROLL for (int l = 0; l < 4; l++) {
          const double coef = (l == leg) ? uk : 0.0;
          QL_UNROLL for (int a = 0; a < 3; a++) v[3 * l + a] += coef * nk[a];
        }
      }
      chol_solve12(L, inv_d, v, dx);
      QL_UNROLL for (int i = 0; i < 12; i++) x[i] += dx[i];
    }
  }
  return res;
}


// ------------------------------------------------------ one control step ---

struct RobotOut {
  double tau[12]; // clamped joint efforts, 0 for non-support legs
  double grf[12]; // QP solution x (ground reaction forces, base frame)
  int status;
  int iters;
  int n_active;
};

// QP Hessian block (leg l, leg m), l >= m:  S_f + X_l' S_t X_m  with X = skew(r)
QL_HD void hessian_block(const double St[3], const double r[3], const double rp[3], double E[9]) {
  const double x = r[0], y = r[1], z = r[2], xp = rp[0], yp = rp[1], zp = rp[2];
  E[0] = St[1] * z * zp + St[2] * y * yp; E[1] = -St[2] * y * xp;                 E[2] = -St[1] * z * xp;
  E[3] = -St[2] * x * yp;                 E[4] = St[0] * z * zp + St[2] * x * xp; E[5] = -St[0] * z * yp;
  E[6] = -St[1] * x * zp;                 E[7] = -St[0] * y * zp;                 E[8] = St[0] * y * yp + St[1] * x * xp;
}

// normals_w: per-leg surface normals in the world frame ([4][3]) when kPerLeg, else ignored.
template <bool kPerLeg, class Scratch>
QL_HD void balance_robot(const DeviceParams &P, const RobotIn &in, const double *normals_w, Scratch &scr,
                         RobotOut &out) {
  double Rm[9], gB[3], b[6];
  quat_to_matrix(in.quat, Rm);
  {
    const double gW[3] = {0.0, 0.0, -P.grav};
    irot(Rm, gW, gB); // VirtualModelController.cpp:165-166, ContactForceDistribution.cpp:518-519
  }
  virtual_wrench(P, in, Rm, gB, b);

  QL_UNROLL for (int i = 0; i < 12; i++) { out.tau[i] = 0.0; out.grf[i] = 0.0; }
  out.status = kStatusOk; out.iters = 0; out.n_active = 0;
  const unsigned stance = in.stance & 0xFu;
  const int nS = (int)((stance & 1u) + ((stance >> 1) & 1u) + ((stance >> 2) & 1u) + ((stance >> 3) & 1u));
  if (nS == 0) return; // ContactForceDistribution.cpp:127-132

  // foot positions of the support legs (prepareOptimization, :189-199)
  double r[4][3];
  QL_UNROLL for (int l = 0; l < 4; l++) {
    r[l][0] = r[l][1] = r[l][2] = 0.0;
    if ((stance >> l) & 1u) {
      LegFrames F;
      leg_frames(P, l, &in.q[3 * l], F);
      r[l][0] = F.p[3][0]; r[l][1] = F.p[3][1]; r[l][2] = F.p[3][2];
    }
  }

  // friction pyramids (addMinimalForceConstraints :223-237, addFrictionConstraints :272-309)
  Pyramid py[4];
  {
    const double ey[3] = {0.0, 1.0, 0.0}, ez[3] = {0.0, 0.0, 1.0};
    double yB[3];
    irot(Rm, ey, yB);
    QL_UNROLL for (int l = 0; l < (kPerLeg ? 4 : 1); l++) {
      double nW[3];
      if (kPerLeg) { nW[0] = normals_w[3 * l]; nW[1] = normals_w[3 * l + 1]; nW[2] = normals_w[3 * l + 2]; }
      else rot(Rm, ez, nW); // ros_balance_controller.cpp:378
      irot(Rm, nW, py[l].n);
      cross3(py[l].n, yB, py[l].t1);
      double nn = 1.0 / sqrt(dot3(py[l].t1, py[l].t1));
      QL_UNROLL for (int a = 0; a < 3; a++) py[l].t1[a] *= nn;
      cross3(py[l].n, py[l].t1, py[l].t2);
      nn = 1.0 / sqrt(dot3(py[l].t2, py[l].t2));
      QL_UNROLL for (int a = 0; a < 3; a++) py[l].t2[a] *= nn;
    }
  }

  // QP data: G = A'SA + W (packed lower), g0 = -A'Sb, legs outside the stance set decoupled
  double G[78], g0[12], x[12];
  {
    const double Sf[3] = {P.S[0], P.S[1], P.S[2]}, St[3] = {P.S[3], P.S[4], P.S[5]};
    const double SF[3] = {Sf[0] * b[0], Sf[1] * b[1], Sf[2] * b[2]};
    const double ST[3] = {St[0] * b[3], St[1] * b[4], St[2] * b[5]};
    QL_UNROLL for (int l = 0; l < 4; l++) {
      const bool on_l = (stance >> l) & 1u;
      // X' y = y x r
      double c[3];
      cross3(ST, r[l], c);
      QL_UNROLL for (int a = 0; a < 3; a++) g0[3 * l + a] = on_l ? -(SF[a] + c[a]) : 0.0;
      QL_UNROLL for (int mm = 0; mm <= l; mm++) {
        const bool on = on_l && ((stance >> mm) & 1u);
        double E[9];
        hessian_block(St, r[l], r[mm], E);
        QL_UNROLL for (int a = 0; a < 3; a++)
          QL_UNROLL for (int bb = 0; bb < 3; bb++) {
            const int i = 3 * l + a, j = 3 * mm + bb;
            if (j > i) continue;
            double v = on ? (E[a * 3 + bb] + (a == bb ? Sf[a] : 0.0)) : 0.0;
            if (i == j) v += P.w_reg;
            G[tri(i, j)] = v;
          }
      }
    }
  }

  const QpResult qr = force_qp_solve<kPerLeg>(G, g0, py, P.mu, P.f_min, stance, nS, scr, x);
  out.status = qr.status; out.iters = qr.iters; out.n_active = qr.n_active;
  if (qr.status != kStatusOk) return; // efforts stay untouched in the reference; we report zeros + status

  // computeJointTorques, ContactForceDistribution.cpp:516-578, then the clamp
  QL_UNROLL for (int l = 0; l < 4; l++) {
    if (!((stance >> l) & 1u)) continue;
    LegFrames F;
    leg_frames(P, l, &in.q[3 * l], F);
    double J[9], Gq[3];
    leg_jac_grav(P, l, F, gB, J, Gq);
    const double fc[3] = {-x[3 * l], -x[3 * l + 1], -x[3 * l + 2]};
    QL_UNROLL for (int j = 0; j < 3; j++) {
      double t = (J[j] * fc[0] + J[3 + j] * fc[1] + J[6 + j] * fc[2]) + Gq[j];
      t = t > P.tau_max ? P.tau_max : t;
      t = t < -P.tau_max ? -P.tau_max : t;
      out.tau[3 * l + j] = t;
      out.grf[3 * l + j] = x[3 * l + j];
    }
  }
}

} // namespace qlamd
