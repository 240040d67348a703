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
// Work split inside a wavefront (see balance_kernel.hip):
//   phase A  one lane per (robot, leg): FK, translation Jacobian, gravity torque
//   phase B  one lane per robot: wrench, QP assembly, Cholesky, active-set QP
//   phase C  one lane per (robot, leg): tau = J'(-f) + G(q), clamp
// Everything a lane indexes with a compile-time constant lives in VGPRs (the
// 12x12 Cholesky factor of the QP Hessian included); arrays that need run-time
// indices and the hand-over between phases go through a `Scratch` object that
// the kernel maps onto LDS with an [element][robot] layout (bank = robot, so
// a per-lane index never conflicts).
//
// Code size matters as much as flop count here: a wavefront runs this once,
// so every instruction byte is fetched cold.  The QP is therefore written as
// a state machine around ONE inlined copy of the 12x12 triangular solves
// (states X0 / G / Z / POLISH / REBUILD differ only in how the right-hand
// side is built and consumed); divergent lanes in different states still
// share the solve's instruction stream.
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
#define QL_NOUNROLL _Pragma("nounroll")
#define QL_HD_CALL __host__ __device__ __attribute__((noinline)) // real calls: keeps cold straight-line code small
#else
#define QL_HD inline
#define QL_UNROLL
#define QL_NOUNROLL
#define QL_HD_CALL inline
#endif

namespace qlamd {

constexpr int kStatusOk = 0, kStatusInfeasible = 1, kStatusNotPd = 2, kStatusMaxIter = 3, kStatusDependentEquality = 5;
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
  // leg chains, 88 doubles per leg: R0[4][9] fixed rotations (row-major), xyz[4][3] joint
  // origins, mass[4], mcom[4][3] = mass * com (link frame), inertia[4][6] about the com
  // (ixx ixy ixz iyy iyz izz, link frame); offsets below
  double legtab[4 * 88];
  int refine_passes;           // lane-cooperative kernel: refinement passes on the final working set
  int keep_on_failure;         // QLAMD_OPT_ON_FAILURE: 1 = leave the efforts / forces of a robot whose solve failed untouched
  int warm_fallback;           // QLAMD_OPT_WARM_FALLBACK: 1 = a robot whose warm start is rejected is solved again, cold, by the same launch
};
constexpr int kTabR0 = 0, kTabXyz = 36, kTabMass = 48, kTabMcom = 52, kTabInertia = 64, kTabPerLeg = 88;

struct RobotIn {
  double q[12];
  double pos[3], quat[4], linvel[3], angvel[3];
  double dpos[3], dquat[4], dlinvel[3], dangvel[3];
  unsigned stance; // bit l = leg l supports
  bool has_wrench;  // true: use `wrench` (computeForceDistribution's arguments) instead of the VMC
  double wrench[6]; // (F_B, T_B)
};

// 1/x and 1/sqrt(x).  Device: hardware seed + two Newton steps (1-2 ulp, ~6 instructions) instead of the ~12-25
// instruction IEEE division / square root sequences; host build (tests): the plain expressions.
#if defined(__HIP_DEVICE_COMPILE__)
QL_HD double ql_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y + y * (1.0 - x * y);
  y = y + y * (1.0 - x * y);
  return y;
}
QL_HD double ql_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y + y * (0.5 - 0.5 * x * y * y);
  y = y + y * (0.5 - 0.5 * x * y * y);
  return y;
}
#else
QL_HD double ql_rcp(double x) { return 1.0 / x; }
QL_HD double ql_rsqrt(double x) { return 1.0 / sqrt(x); }
#endif

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

// ---------------------------------------------------- analytic leg IK ----
// QuadrupedKinematics::InverseKinematicsSolve (quadrupedkinematics.cpp:377-483): foot position in the base frame
// -> (q1, q2, q3) of one leg.  The hip frame is the first joint's origin (setHipPoseInBase, :109-122) = R0[0],
// xyz[0] of the leg table.  config = the row of `results` the reference returns: 0 OUT_LEFT, 1 IN_RIGHT,
// 2 IN_LEFT, 3 OUT_RIGHT (:466-473).  geom = {d, l1, l2}.  Returns false when an angle is NaN (:478-483).
QL_HD double map_to_pi(double q) { // :554-563, as written there (2 pi - q above pi)
  const double pi = 3.14159265358979323846;
  double out = q;
  if (q > pi) out = 2 * pi - q;
  if (q < -pi) out = 2 * pi + q;
  return out;
}

template <class Tab>
QL_HD bool leg_inverse_kinematics(const Tab &tab, const double p_base[3], int config, const double geom[3], double q[3]) {
  const double pi = 3.14159265358979323846;
  const double d = geom[0], l1 = geom[1], l2 = geom[2];
  double rel[3], ph[3];
  QL_UNROLL for (int i = 0; i < 3; i++) rel[i] = p_base[i] - tab[kTabXyz + i];
  QL_UNROLL for (int i = 0; i < 3; i++)
    ph[i] = tab[kTabR0 + i] * rel[0] + tab[kTabR0 + 3 + i] * rel[1] + tab[kTabR0 + 6 + i] * rel[2];
  const double px = ph[0], py = ph[1], pz = ph[2];
  double cos_theta3 = (l2 * l2 + l1 * l1 - ((px * px + py * py + pz * pz) - d * d)) / 2 / l1 / l2;
  if (cos_theta3 < -1) cos_theta3 = -1;
  if (cos_theta3 > 1) cos_theta3 = 1;
  const double theta3 = (config < 2) ? pi - acos(cos_theta3) : -pi + acos(cos_theta3);
  const double alpha = atan2(py, px);
  const double rxy = sqrt(fabs(px * px + py * py - d * d));
  const double q3 = map_to_pi(theta3);
  const double b = atan2(l2 * sin(q3), l1 + l2 * cos(q3));
  double q1, q2;
  if ((config & 1) == 0) { // left arm (:416-429); the reference writes nothing when a == 0
    q1 = map_to_pi(alpha - atan2(d, rxy));
    const double a = atan2(pz, -rxy);
    q2 = a > 0 ? map_to_pi(a - b - pi) : a < 0 ? map_to_pi(a - b + pi) : (double)NAN;
  } else {                 // right arm (:431-435)
    q1 = map_to_pi(alpha + atan2(-d, -rxy));
    const double a = atan2(pz, rxy);
    q2 = map_to_pi(a - b + pi);
  }
  q[0] = q1; q[1] = q2; q[2] = q3;
  return !(q1 != q1) && !(q2 != q2) && !(q3 != q3);
}

// ------------------------------------------------------- leg kinematics ----

// Scratch element space (doubles per robot), [element][robot] in LDS:
//   C    91  packed lower Cholesky factor of the active-set Schur complement (12 rows) plus
//            one more row for the candidate's tentative Schur row (q can reach 12)
//   U    13  multipliers of the active constraints (+1 for the candidate)
//   R    13  dual step direction
//   UO   13  multipliers saved at the start of an outer iteration
//   XO   12  x saved at the start of an outer iteration
//   X0   12  unconstrained minimiser (final polish), then the final x for phase C
//   FEET 12  foot positions in base (phase A -> B)
//   JAC  36  translation Jacobians, row-major 3x3 per leg (phase A -> C)
//   GQ   12  gravity torques (phase A -> C)
//   STATUS 1 per-robot status as a double (phase B -> C)
constexpr int kScrC = 0, kScrU = 91, kScrR = 104, kScrUo = 117, kScrXo = 130, kScrX0 = 142, kScrFeet = 154,
              kScrJac = 166, kScrGq = 202, kScrStatus = 214, kScratchDoubles = 215;

struct HostScratch {
  double a[kScratchDoubles];
  QL_HD double &at(int e) { return a[e]; }
};

// FK + Jacobian + gravity torque of ONE leg (phase A).  `tab` = this leg's 64-double
// block of DeviceParams::legtab (an LDS copy in the kernel: the leg differs per lane).
//   T_seg(q) = Trans(xyz) R0 Rz(q)          (kdl_parser placement)
//   J[:,i]   = z_i x (p_foot - p_i)         (ChainJntToJacSolver, quadrupedkinematics.cpp:214-278)
//   G_i      = -g . (z_i x sum_{k>=i} m_k (c_k - p_i))   (ChainDynParam::JntToGravity, :485-552)
template <class Tab>
QL_HD void leg_kinematics(const Tab &tab, const double q[3], const double g[3], double foot[3], double J[9],
                          double Gq[3]) {
  double Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double pc[3] = {0, 0, 0};
  double z[3][3], p[3][3], mc[3] = {0, 0, 0}; // joint axes, joint origins
  double H[4][3];                              // m_k * c_k per link, base frame
  QL_UNROLL for (int k = 0; k < 4; k++) {
    double R0[9], t[3], mcom[3];
    QL_UNROLL for (int i = 0; i < 9; i++) R0[i] = tab[kTabR0 + 9 * k + i];
    QL_UNROLL for (int i = 0; i < 3; i++) { t[i] = tab[kTabXyz + 3 * k + i]; mcom[i] = tab[kTabMcom + 3 * k + i]; }
    const double m = tab[kTabMass + k];
    double Rs[9];
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
    QL_UNROLL for (int i = 0; i < 9; i++) Rc[i] = Rn[i];
    QL_UNROLL for (int i = 0; i < 3; i++) {
      pc[i] = pn[i];
      H[k][i] = m * pn[i] + (Rn[i * 3] * mcom[0] + Rn[i * 3 + 1] * mcom[1] + Rn[i * 3 + 2] * mcom[2]);
    }
    if (k < 3) {
      QL_UNROLL for (int i = 0; i < 3; i++) { z[k][i] = Rn[i * 3 + 2]; p[k][i] = pn[i]; }
    }
  }
  (void)mc;
  foot[0] = pc[0]; foot[1] = pc[1]; foot[2] = pc[2];
  double Hs[3] = {H[3][0], H[3][1], H[3][2]};
  double M = tab[kTabMass + 3];
  QL_UNROLL for (int i = 2; i >= 0; i--) {
    QL_UNROLL for (int a = 0; a < 3; a++) Hs[a] += H[i][a];
    M += tab[kTabMass + i];
    const double d[3] = {pc[0] - p[i][0], pc[1] - p[i][1], pc[2] - p[i][2]};
    double v[3];
    cross3(z[i], d, v);
    J[0 * 3 + i] = v[0]; J[1 * 3 + i] = v[1]; J[2 * 3 + i] = v[2];
    const double h[3] = {Hs[0] - M * p[i][0], Hs[1] - M * p[i][1], Hs[2] - M * p[i][2]};
    double zh[3];
    cross3(z[i], h, zh);
    Gq[i] = -dot3(g, zh);
  }
}

// Phase A for one (robot, leg): results into the robot's scratch.
template <class Tab, class Scratch>
QL_HD void phase_a_leg(const Tab &tab, int leg, bool support, const double q[3], const double quat[4], double grav,
                       Scratch &scr) {
  double foot[3] = {0, 0, 0}, J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Gq[3] = {0, 0, 0};
  if (support) {
    double Rm[9], gB[3];
    quat_to_matrix(quat, Rm);
    const double gW[3] = {0.0, 0.0, -grav};
    irot(Rm, gW, gB); // ContactForceDistribution.cpp:518-519
    leg_kinematics(tab, q, gB, foot, J, Gq);
  }
  QL_UNROLL for (int a = 0; a < 3; a++) {
    scr.at(kScrFeet + 3 * leg + a) = foot[a];
    scr.at(kScrGq + 3 * leg + a) = Gq[a];
  }
  QL_UNROLL for (int a = 0; a < 9; a++) scr.at(kScrJac + 9 * leg + a) = J[a];
}

// Phase C for one (robot, leg): tau = J'(-x_leg) + G(q), clamped
// (ContactForceDistribution.cpp:516-578, ros_balance_controller.cpp:451-454).
template <class Scratch>
QL_HD void phase_c_leg(int leg, bool live, double tau_max, Scratch &scr, double tau[3], double grf[3]) {
  QL_UNROLL for (int j = 0; j < 3; j++) { tau[j] = 0.0; grf[j] = 0.0; }
  if (!live) return;
  double x[3], J[9];
  QL_UNROLL for (int a = 0; a < 3; a++) x[a] = scr.at(kScrX0 + 3 * leg + a);
  QL_UNROLL for (int a = 0; a < 9; a++) J[a] = scr.at(kScrJac + 9 * leg + a);
  QL_UNROLL for (int j = 0; j < 3; j++) {
    double t = (J[j] * -x[0] + J[3 + j] * -x[1] + J[6 + j] * -x[2]) + scr.at(kScrGq + 3 * leg + j);
    t = t > tau_max ? tau_max : t;
    t = t < -tau_max ? -tau_max : t;
    tau[j] = t;
    grf[j] = x[j];
  }
}

// -------------------------------------------------------------- the QP -----

QL_HD constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// Constraint ids: id = 5*leg + t;  t = 0: n.f >= f_min;  t = 1..4: (mu n +- t1/t2).f >= 0
// in the reference's row order +t1, -t1, +t2, -t2 (ContactForceDistribution.cpp:314-325).
QL_HD int id_leg(int id) { return (id * 13) >> 6; } // id / 5 for id < 20

struct Pyramid { // friction pyramid directions in the base frame
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

QL_HD void leg_part(const double v[12], int leg, double o[3]) {
  QL_UNROLL for (int a = 0; a < 3; a++) o[a] = leg == 0 ? v[a] : leg == 1 ? v[3 + a] : leg == 2 ? v[6 + a] : v[9 + a];
}

// v (12) (+)= coef * nv placed in leg block `leg`
QL_HD void leg_axpy(double v[12], int leg, double coef, const double nv[3]) {
  QL_UNROLL for (int l = 0; l < 4; l++) {
    const double c = (l == leg) ? coef : 0.0;
    QL_UNROLL for (int a = 0; a < 3; a++) v[3 * l + a] += c * nv[a];
  }
}

struct QpResult {
  int status;
  int iters;
  int n_active;
};

// Solve  min 1/2 x'Gx + g0'x  s.t. the pyramid / minimum-force constraints of the stance
// legs.  G: packed lower 12x12 (legs outside `stance` decoupled as w*I), overwritten by its
// Cholesky factor with the RECIPROCAL diagonal stored on the diagonal.  The final x is left
// in scratch [kScrX0, kScrX0+12).
template <bool kPerLeg, class Scratch>
QL_HD QpResult force_qp_solve(double G[78], const double g0[12], const Pyramid (&py)[4], double mu, double f_min,
                              unsigned stance, int nS, Scratch &scr) {
  QpResult res = {kStatusOk, 0, 0};
  const double eps = 2.220446049250313e-16;
  const double inf = INFINITY;

  // --- Cholesky G = L L' (in place); c1 = trace(G), c2 = trace(L^-1) over the stance block
  double c1 = 0.0, c2 = 0.0;
  bool not_pd = false;
  QL_UNROLL for (int j = 0; j < 12; j++) {
    const bool on = (stance >> (j / 3)) & 1u;
    if (on) c1 += G[tri(j, j)];
    double d = G[tri(j, j)];
    QL_UNROLL for (int k = 0; k < j; k++) d -= G[tri(j, k)] * G[tri(j, k)];
    if (!(d > 0.0)) not_pd = true;
    const double inv = 1.0 / sqrt(d);
    G[tri(j, j)] = inv;
    if (on) c2 += inv;
    QL_UNROLL for (int i = j + 1; i < 12; i++) {
      double acc = G[tri(i, j)];
      QL_UNROLL for (int k = 0; k < j; k++) acc -= G[tri(i, k)] * G[tri(j, k)];
      G[tri(i, j)] = acc * inv;
    }
  }
  if (not_pd) { res.status = kStatusNotPd; return res; }
  const double *L = G;

  const int m = 5 * nS;
  const double psi_tol = (double)m * eps * c1 * c2 * 100.0; // QuadProg++.cc:246
  uint64_t act = 0, act_old = 0; // active ids, 5 bits each, position k at bits [5k, 5k+5)
  unsigned act_mask = 0, excl = 0;
  int q = 0, q_old = 0;
  double rnorm = 1.0;

  const auto leg_pyr = [&](int leg) -> Pyramid { return pick_pyramid<kPerLeg>(py, leg); };
  const auto act_id = [&](int k) -> int { return (int)((act >> (5 * k)) & 31u); };
  const auto slack = [&](int id, const double xx[12]) -> double {
    const int leg = id_leg(id), t = id - 5 * leg;
    double xl[3], nv[3];
    leg_part(xx, leg, xl);
    constraint_normal(leg_pyr(leg), mu, t, nv);
    return dot3(nv, xl) - (t == 0 ? f_min : 0.0);
  };

  enum { kX0, kSelect, kG, kZ, kPolish, kRebuild, kDone };
  int state = kX0;
  bool fresh = true;        // kSelect: start of an outer iteration (label l1) vs. re-selection (l2)
  int ip = -1, pleg = 0, pt = 0, pass = 0, kk = 0;
  double x[12], v[12], npv[3] = {0, 0, 0};
  double sp = 0.0, ss = 0.0, dd = 0.0;
  QL_UNROLL for (int i = 0; i < 12; i++) { x[i] = 0.0; v[i] = 0.0; }

  for (int tick = 0; tick < 40 * kMaxOuter; tick++) {
    // ------------------------------------------------ pre: choose a candidate
    if (state == kSelect) {
      if (fresh) { res.iters++; excl = 0; }
      double psi = 0.0;
      ss = 0.0; ip = -1;
      const unsigned blocked = act_mask | excl;
      // reference row order: all minimum-force rows, then the friction rows leg by leg
      QL_UNROLL for (int ps = 0; ps < 2; ps++) {
        QL_UNROLL for (int leg = 0; leg < 4; leg++) {
          if (!((stance >> leg) & 1u)) continue;
          const Pyramid &pl = py[kPerLeg ? leg : 0];
          const double dn = pl.n[0] * x[3 * leg] + pl.n[1] * x[3 * leg + 1] + pl.n[2] * x[3 * leg + 2];
          if (ps == 0) {
            const double s = dn - f_min;
            psi += fmin(0.0, s);
            if (s < ss && !((blocked >> (5 * leg)) & 1u)) { ss = s; ip = 5 * leg; }
          } else {
            const double d1 = pl.t1[0] * x[3 * leg] + pl.t1[1] * x[3 * leg + 1] + pl.t1[2] * x[3 * leg + 2];
            const double d2 = pl.t2[0] * x[3 * leg] + pl.t2[1] * x[3 * leg + 1] + pl.t2[2] * x[3 * leg + 2];
            const double sv[4] = {mu * dn + d1, mu * dn - d1, mu * dn + d2, mu * dn - d2};
            QL_UNROLL for (int t = 1; t <= 4; t++) {
              const double s = sv[t - 1];
              psi += fmin(0.0, s);
              if (s < ss && !((blocked >> (5 * leg + t)) & 1u)) { ss = s; ip = 5 * leg + t; }
            }
          }
        }
      }
      const bool feasible = fresh && (fabs(psi) <= psi_tol); // QuadProg++.cc:246-250
      if (feasible || ip < 0 || res.iters > kMaxOuter) {      // :271-274
        if (res.iters > kMaxOuter) res.status = kStatusMaxIter;
        state = (q > 0 && res.status == kStatusOk) ? kPolish : kDone;
        pass = 0;
        if (state == kDone) {
          QL_UNROLL for (int i = 0; i < 12; i++) scr.at(kScrX0 + i) = x[i];
        }
      } else {
        if (fresh) { // save for the degenerate rollback, QuadProg++.cc:252-260
          act_old = act; q_old = q;
          for (int k = 0; k < q; k++) scr.at(kScrUo + k) = scr.at(kScrU + k);
          QL_UNROLL for (int i = 0; i < 12; i++) scr.at(kScrXo + i) = x[i];
        }
        pleg = id_leg(ip); pt = ip - 5 * pleg;
        constraint_normal(leg_pyr(pleg), mu, pt, npv);
        sp = ss;
        scr.at(kScrU + q) = 0.0;
        state = kG;
      }
    }
    if (state == kDone) break;

    // ------------------------------------------------ right-hand side by state
    double rhs[12];
    if (state == kX0) {
      QL_UNROLL for (int i = 0; i < 12; i++) rhs[i] = -g0[i];
    } else if (state == kG) {
      QL_UNROLL for (int i = 0; i < 12; i++) rhs[i] = 0.0;
      leg_axpy(rhs, pleg, 1.0, npv);
    } else if (state == kZ) {
      QL_UNROLL for (int i = 0; i < 12; i++) rhs[i] = v[i];
    } else if (state == kPolish) {
      // u = M^-1 (b_A - N'x), rhs = N u      (corrected semi-normal equations)
      if (pass == 0) {
        QL_UNROLL for (int i = 0; i < 12; i++) x[i] = scr.at(kScrX0 + i);
      }
      QL_UNROLL for (int i = 0; i < 12; i++) rhs[i] = 0.0;
      for (int k = 0; k < q; k++) {
        double acc = -slack(act_id(k), x);
        for (int j = 0; j < k; j++) acc -= scr.at(kScrC + tri(k, j)) * scr.at(kScrR + j);
        scr.at(kScrR + k) = acc / scr.at(kScrC + tri(k, k));
      }
      for (int k = q - 1; k >= 0; k--) {
        double acc = scr.at(kScrR + k);
        for (int j = k + 1; j < q; j++) acc -= scr.at(kScrC + tri(j, k)) * scr.at(kScrR + j);
        const double uk = acc / scr.at(kScrC + tri(k, k));
        scr.at(kScrR + k) = uk;
        const int id = act_id(k), leg = id_leg(id);
        double nk[3];
        constraint_normal(leg_pyr(leg), mu, id - 5 * leg, nk);
        leg_axpy(rhs, leg, uk, nk);
      }
    } else { // kRebuild: row kk of the Schur factor of the restored working set
      const int id = act_id(kk), leg = id_leg(id);
      double nk[3];
      constraint_normal(leg_pyr(leg), mu, id - 5 * leg, nk);
      QL_UNROLL for (int i = 0; i < 12; i++) rhs[i] = 0.0;
      leg_axpy(rhs, leg, 1.0, nk);
    }

    // ------------------------------------------------ sol = G^-1 rhs  (the one copy)
    double sol[12];
    {
      double y[12];
      QL_UNROLL for (int i = 0; i < 12; i++) {
        double acc = rhs[i];
        QL_UNROLL for (int j = 0; j < i; j++) acc -= L[tri(i, j)] * y[j];
        y[i] = acc * L[tri(i, i)];
      }
      QL_UNROLL for (int i = 11; i >= 0; i--) {
        double acc = y[i];
        QL_UNROLL for (int j = i + 1; j < 12; j++) acc -= L[tri(j, i)] * sol[j];
        sol[i] = acc * L[tri(i, i)];
      }
    }

    // ------------------------------------------------ consume by state
    bool have_z = false;
    if (state == kX0) {
      QL_UNROLL for (int i = 0; i < 12; i++) { x[i] = sol[i]; scr.at(kScrX0 + i) = sol[i]; }
      state = kSelect; fresh = true;
    } else if (state == kPolish) {
      QL_UNROLL for (int i = 0; i < 12; i++) x[i] += sol[i];
      if (++pass == 2) {
        QL_UNROLL for (int i = 0; i < 12; i++) scr.at(kScrX0 + i) = x[i];
        state = kDone;
      }
    } else if (state == kRebuild) {
      double d2 = 0.0;
      for (int k = 0; k <= kk; k++) {
        const int id = act_id(k), leg = id_leg(id);
        double nk[3], gk[3];
        constraint_normal(leg_pyr(leg), mu, id - 5 * leg, nk);
        leg_part(sol, leg, gk);
        double acc = dot3(nk, gk);
        for (int j = 0; j < k; j++) acc -= scr.at(kScrC + tri(k, j)) * scr.at(kScrC + tri(kk, j));
        if (k < kk) scr.at(kScrC + tri(kk, k)) = acc / scr.at(kScrC + tri(k, k));
        else d2 = acc;
      }
      scr.at(kScrC + tri(kk, kk)) = sqrt(fmax(d2, 0.0));
      if (++kk >= q) { state = kSelect; fresh = false; }
    } else if (state == kG) {
      // g = sol.  Tentative Schur row y = C^-1 N'g into row q of C, dd = n_p'g - y'y,
      // r = C^-T y, v = n_p - N r
      double gl[3];
      leg_part(sol, pleg, gl);
      dd = dot3(npv, gl);
      for (int k = 0; k < q; k++) {
        const int id = act_id(k), leg = id_leg(id);
        double nk[3], gk[3];
        constraint_normal(leg_pyr(leg), mu, id - 5 * leg, nk);
        leg_part(sol, leg, gk);
        double acc = dot3(nk, gk);
        for (int j = 0; j < k; j++) acc -= scr.at(kScrC + tri(k, j)) * scr.at(kScrC + tri(q, j));
        const double yk = acc / scr.at(kScrC + tri(k, k));
        scr.at(kScrC + tri(q, k)) = yk;
        dd -= yk * yk;
      }
      if (q > 0) {
        QL_UNROLL for (int i = 0; i < 12; i++) v[i] = 0.0;
        leg_axpy(v, pleg, 1.0, npv);
        for (int k = q - 1; k >= 0; k--) {
          double acc = scr.at(kScrC + tri(q, k));
          for (int j = k + 1; j < q; j++) acc -= scr.at(kScrC + tri(j, k)) * scr.at(kScrR + j);
          const double rk = acc / scr.at(kScrC + tri(k, k));
          scr.at(kScrR + k) = rk;
          const int id = act_id(k), leg = id_leg(id);
          double nk[3];
          constraint_normal(leg_pyr(leg), mu, id - 5 * leg, nk);
          leg_axpy(v, leg, -rk, nk);
        }
        state = kZ;
      } else {
        have_z = true; // z = g
      }
    } else { // kZ
      have_z = true;
    }

    // ------------------------------------------------ step (QuadProg++.cc:304-445), z = sol
    if (have_z) {
      double zl[3];
      leg_part(sol, pleg, zl);
      const double zn = dot3(zl, npv);
      double zz = 0.0;
      QL_UNROLL for (int i = 0; i < 12; i++) zz += sol[i] * sol[i];
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
      if (t >= inf) { res.status = kStatusInfeasible; state = kDone; break; } // :339-344
      const bool dual_only = (t2 >= inf);
      if (!dual_only) {
        QL_UNROLL for (int i = 0; i < 12; i++) x[i] += t * sol[i];
      }
      for (int k = 0; k < q; k++) scr.at(kScrU + k) -= t * scr.at(kScrR + k);
      scr.at(kScrU + q) += t;

      if (!dual_only && fabs(t - t2) < eps) {
        // full step: ip becomes active (:384-421); its Schur row is already in row q of C
        const double delta = dd > 0.0 ? sqrt(dd) : 0.0;
        if (delta <= eps * rnorm) {
          // degenerate: exclude ip, restore the state saved at the start of the outer iteration
          excl |= 1u << ip;
          act = act_old; q = q_old;
          act_mask = 0;
          for (int k = 0; k < q; k++) {
            act_mask |= 1u << act_id(k);
            scr.at(kScrU + k) = scr.at(kScrUo + k);
          }
          QL_UNROLL for (int i = 0; i < 12; i++) x[i] = scr.at(kScrXo + i);
          kk = 0;
          if (q > 0) state = kRebuild; else { state = kSelect; fresh = false; }
        } else {
          scr.at(kScrC + tri(q, q)) = delta;
          rnorm = fmax(rnorm, delta);
          act |= (uint64_t)ip << (5 * q);
          act_mask |= 1u << ip;
          q++;
          state = kSelect; fresh = true;
        }
      } else {
        // partial or dual-only step: drop the blocking constraint at position lpos (:346-362, :423-445).
        // Remove row lpos of C (C C' = N'G^-1 N) and restore the triangle, working in the OLD index
        // space: for the column pair (c, c+1) a rotation computed from old row c+1 zeroes its
        // overhang, rows below are rotated, finally the rows move up by one.
        for (int c = lpos; c < q - 1; c++) {
          const double a0 = scr.at(kScrC + tri(c + 1, c));
          const double b0 = scr.at(kScrC + tri(c + 1, c + 1));
          const double h = sqrt(a0 * a0 + b0 * b0);
          const double cs = h > 0.0 ? a0 / h : 1.0, sn = h > 0.0 ? b0 / h : 0.0;
          for (int i = c + 2; i < q; i++) {
            const double a = scr.at(kScrC + tri(i, c));
            const double b = scr.at(kScrC + tri(i, c + 1));
            scr.at(kScrC + tri(i, c)) = cs * a + sn * b;
            scr.at(kScrC + tri(i, c + 1)) = -sn * a + cs * b;
          }
          scr.at(kScrC + tri(c + 1, c)) = h;
          scr.at(kScrC + tri(c + 1, c + 1)) = 0.0;
        }
        for (int i = lpos + 1; i < q; i++)
          for (int j = 0; j < i; j++) scr.at(kScrC + tri(i - 1, j)) = scr.at(kScrC + tri(i, j));
        act_mask &= ~(1u << act_id(lpos));
        for (int k = lpos; k < q; k++) scr.at(kScrU + k) = scr.at(kScrU + k + 1);
        {
          const uint64_t low = act & ((1ull << (5 * lpos)) - 1ull);
          const uint64_t high = (act >> (5 * (lpos + 1))) << (5 * lpos);
          act = low | high;
        }
        q--;
        if (!dual_only) sp = slack(ip, x); // :436-440
        state = kG;
      }
    }
  }
  if (state != kDone && res.status == kStatusOk) res.status = kStatusMaxIter;
  res.n_active = q;
  return res;
}

// ------------------------------------------------------ phase B: one robot ---

// QP Hessian block (leg l, leg m):  X_l' S_t X_m  with X = skew(r)
QL_HD void hessian_block(const double St[3], const double r[3], const double rp[3], double E[9]) {
  const double x = r[0], y = r[1], z = r[2], xp = rp[0], yp = rp[1], zp = rp[2];
  E[0] = St[1] * z * zp + St[2] * y * yp; E[1] = -St[2] * y * xp;                 E[2] = -St[1] * z * xp;
  E[3] = -St[2] * x * yp;                 E[4] = St[0] * z * zp + St[2] * x * xp; E[5] = -St[0] * z * yp;
  E[6] = -St[1] * x * zp;                 E[7] = -St[0] * y * zp;                 E[8] = St[0] * y * yp + St[1] * x * xp;
}

// Wrench, pyramids, QP assembly and solve.  Foot positions come from scratch (phase A); the
// solution x is left in scratch [kScrX0, +12) for phase C.
// normals_w: per-leg surface normals in the world frame ([4][3]) when kPerLeg, else ignored.
template <bool kPerLeg, class Scratch>
QL_HD QpResult phase_b_robot(const DeviceParams &P, const RobotIn &in, const double *normals_w, Scratch &scr) {
  QpResult none = {kStatusOk, 0, 0};
  const unsigned stance = in.stance & 0xFu;
  const int nS = (int)((stance & 1u) + ((stance >> 1) & 1u) + ((stance >> 2) & 1u) + ((stance >> 3) & 1u));
  if (nS == 0) { // ContactForceDistribution.cpp:127-132
    QL_UNROLL for (int i = 0; i < 12; i++) scr.at(kScrX0 + i) = 0.0;
    return none;
  }
  double Rm[9], gB[3], b[6];
  quat_to_matrix(in.quat, Rm);
  {
    const double gW[3] = {0.0, 0.0, -P.grav};
    irot(Rm, gW, gB); // VirtualModelController.cpp:165-166
  }
  if (in.has_wrench) {
    QL_UNROLL for (int k = 0; k < 6; k++) b[k] = in.wrench[k];
  } else {
    virtual_wrench(P, in, Rm, gB, b);
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

  // QP data: G = A'SA + W (packed lower), g0 = -A'Sb  (prepareOptimization :168-206, objective :388)
  double G[78], g0[12];
  {
    double r[4][3];
    QL_UNROLL for (int l = 0; l < 4; l++)
      QL_UNROLL for (int a = 0; a < 3; a++) r[l][a] = scr.at(kScrFeet + 3 * l + a);
    const double Sf[3] = {P.S[0], P.S[1], P.S[2]}, St[3] = {P.S[3], P.S[4], P.S[5]};
    const double SF[3] = {Sf[0] * b[0], Sf[1] * b[1], Sf[2] * b[2]};
    const double ST[3] = {St[0] * b[3], St[1] * b[4], St[2] * b[5]};
    QL_UNROLL for (int l = 0; l < 4; l++) {
      const bool on_l = (stance >> l) & 1u;
      double c[3];
      cross3(ST, r[l], c); // X' y = y x r
      QL_UNROLL for (int a = 0; a < 3; a++) g0[3 * l + a] = on_l ? -(SF[a] + c[a]) : 0.0;
      QL_UNROLL for (int mm = 0; mm <= l; mm++) {
        const bool on = on_l && ((stance >> mm) & 1u);
        double E[9];
        hessian_block(St, r[l], r[mm], E);
        QL_UNROLL for (int a = 0; a < 3; a++)
          QL_UNROLL for (int bb = 0; bb < 3; bb++) {
            const int i = 3 * l + a, j = 3 * mm + bb;
            if (j > i) continue;
            double val = on ? (E[a * 3 + bb] + (a == bb ? Sf[a] : 0.0)) : 0.0;
            if (i == j) val += P.w_reg;
            G[tri(i, j)] = val;
          }
      }
    }
  }
  return force_qp_solve<kPerLeg>(G, g0, py, P.mu, P.f_min, stance, nS, scr);
}

} // namespace qlamd
