// Floating-base dynamics of the whole 18-DoF robot and the whole-body force/torque QP, lane-cooperative:
// 16 lanes per robot, 4 robots per wavefront.  Device-only.
//
// SURVEY.md section 8 row f4 -- what BASELINE.json's north_star describes ("batched articulated-body/CRBA pass for
// the 18-DoF tree ... feeding a batched dense active-set QP (~24 vars, friction-cone + torque-limit constraints)")
// and the reference does NOT contain: there is no reference file to cite.  The algorithms are
// Featherstone's (Rigid Body Dynamics Algorithms, 2008): recursive Newton-Euler and the composite-rigid-body
// algorithm for a floating base (Tables 5.1, 9.6, section 9.4), written here in BASE coordinates -- every spatial
// vector and inertia is expressed in the base frame, so the recursions along a leg become prefix / suffix sums over
// the four lanes of a quad and the base terms become sums over the row.  (The CPU oracle, oracle/oracle_wholebody.c,
// uses the link-coordinate form with Pluecker transforms: two formulations, one answer.)
//
// Lane 4*leg + k: body k of the leg -- k = 0,1,2 the links behind the joints HAA, HFE, KFE, k = 3 the fixed foot link.
// A rigid-body inertia about the base origin is 10 numbers: m, h = m c, I (xx xy xz yy yz zz).
// Spatial vectors are [angular ; linear]; the interface order is [linear ; angular] like the reference's (F, T).
#pragma once

#include "balance_coop.hpp"

namespace qlamd {
namespace coop {

struct WbParamsDev {
  double base_m, base_h[3], base_I[6]; // base_link about the base origin
  double w_tau, tau_max, grav;
};

struct WbInertia { double m, h[3], I[6]; };

struct WbLink {
  double z[3], p[3]; // joint axis and joint origin of my body's frame, base coordinates (z unused on the foot lane)
  double pf[3];      // foot-frame origin of my leg
  WbInertia X;       // my body
};

// sum over the lanes c' >= c of my quad / over c' <= c
__device__ __forceinline__ double quad_suffix(double x, int c) {
  double y = x + sel(c < 3, dpp<0xF9>(x), 0.0); // quad_perm [1,2,3,3]
  y += sel(c < 2, dpp<0xFE>(y), 0.0);           // quad_perm [2,3,3,3]
  return y;
}
__device__ __forceinline__ double quad_prefix(double x, int c) {
  double y = x + sel(c >= 1, dpp<0x90>(x), 0.0); // quad_perm [0,0,1,2]
  y += sel(c >= 2, dpp<0x40>(y), 0.0);           // quad_perm [0,0,0,1]
  return y;
}

// sum over the four legs of the value held by the lanes c == 0 (the legs' root bodies), replicated on all 16 lanes:
// two rotations add the lanes of equal c, the quad broadcast then picks c == 0 -- no masking needed
__device__ __forceinline__ double legs_root_sum(double x) {
  x += dpp<0x128>(x); // row_ror:8
  x += dpp<0x124>(x); // row_ror:4
  return quad_bc<0>(x);
}

__device__ __forceinline__ void sym_mul(const double I[6], const double v[3], double o[3]) {
  o[0] = I[0] * v[0] + I[1] * v[1] + I[2] * v[2];
  o[1] = I[1] * v[0] + I[3] * v[1] + I[4] * v[2];
  o[2] = I[2] * v[0] + I[4] * v[1] + I[5] * v[2];
}
// [n ; f] = X [w ; v]
__device__ __forceinline__ void inertia_mul(const WbInertia &X, const double w[3], const double v[3], double n[3], double f[3]) {
  double Iw[3], hv[3], hw[3];
  sym_mul(X.I, w, Iw);
  cross3(X.h, v, hv);
  cross3(X.h, w, hw);
#pragma unroll
  for (int a = 0; a < 3; a++) { n[a] = Iw[a] + hv[a]; f[a] = X.m * v[a] - hw[a]; }
}

// Inertia of body k of a leg about the base origin, in base coordinates, from its frame (Rc, pc): mass, m * com and the
// inertia about the com in the body frame come from the model table.
__device__ __forceinline__ void wb_body_inertia(const CoopTab &tab, int k, const double Rc[9], const double pc[3], WbInertia &X) {
  const double m = tab[kTabMass + k];
  double mc[3], Ic[6];
#pragma unroll
  for (int a = 0; a < 3; a++) mc[a] = tab[kTabMcom + 3 * k + a];
#pragma unroll
  for (int a = 0; a < 6; a++) Ic[a] = tab[kTabInertia + 6 * k + a];
  X.m = m;
#pragma unroll
  for (int a = 0; a < 3; a++) X.h[a] = m * pc[a] + (Rc[3 * a] * mc[0] + Rc[3 * a + 1] * mc[1] + Rc[3 * a + 2] * mc[2]);
  double T[9]; // Rc * Ic
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double r0 = Rc[3 * a], r1 = Rc[3 * a + 1], r2 = Rc[3 * a + 2];
    T[3 * a + 0] = r0 * Ic[0] + r1 * Ic[1] + r2 * Ic[2];
    T[3 * a + 1] = r0 * Ic[1] + r1 * Ic[3] + r2 * Ic[4];
    T[3 * a + 2] = r0 * Ic[2] + r1 * Ic[4] + r2 * Ic[5];
  }
  const double im = m > 0.0 ? rcp_nr(m) : 0.0;
  const double hh = (X.h[0] * X.h[0] + X.h[1] * X.h[1] + X.h[2] * X.h[2]) * im; // m |c|^2
  const auto rot_entry = [&](int a, int b) { return T[3 * a] * Rc[3 * b] + T[3 * a + 1] * Rc[3 * b + 1] + T[3 * a + 2] * Rc[3 * b + 2]; };
  X.I[0] = rot_entry(0, 0) + hh - X.h[0] * X.h[0] * im;
  X.I[1] = rot_entry(0, 1) - X.h[0] * X.h[1] * im;
  X.I[2] = rot_entry(0, 2) - X.h[0] * X.h[2] * im;
  X.I[3] = rot_entry(1, 1) + hh - X.h[1] * X.h[1] * im;
  X.I[4] = rot_entry(1, 2) - X.h[1] * X.h[2] * im;
  X.I[5] = rot_entry(2, 2) + hh - X.h[2] * X.h[2] * im;
}

// Frames of the chain up to my body, my body's inertia about the base origin.  tab = my leg's block of the model table
// (LDS), (sj, cj) = sine / cosine of MY joint angle (anything on the foot lane).
__device__ __forceinline__ void wb_link(const CoopTab &tab, int c, double sj, double cj, WbLink &L) {
  double Rc[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0}, pc[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double R0[9], t[3], Rs[9];
#pragma unroll
    for (int a = 0; a < 9; a++) R0[a] = tab[kTabR0 + 9 * j + a];
#pragma unroll
    for (int a = 0; a < 3; a++) t[a] = tab[kTabXyz + 3 * j + a];
    if (j < 3) {
      const double sk = j == 0 ? quad_bc<0>(sj) : j == 1 ? quad_bc<1>(sj) : quad_bc<2>(sj);
      const double ck = j == 0 ? quad_bc<0>(cj) : j == 1 ? quad_bc<1>(cj) : quad_bc<2>(cj);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        Rs[a * 3 + 0] = R0[a * 3 + 0] * ck + R0[a * 3 + 1] * sk;
        Rs[a * 3 + 1] = R0[a * 3 + 1] * ck - R0[a * 3 + 0] * sk;
        Rs[a * 3 + 2] = R0[a * 3 + 2];
      }
    } else {
#pragma unroll
      for (int a = 0; a < 9; a++) Rs[a] = R0[a];
    }
    const bool act = j <= c; // my frame is the product of the first c + 1 segments
    double pn[3], Rn[9];
#pragma unroll
    for (int a = 0; a < 3; a++) pn[a] = pc[a] + (Rc[3 * a] * t[0] + Rc[3 * a + 1] * t[1] + Rc[3 * a + 2] * t[2]);
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) Rn[3 * a + b] = Rc[3 * a] * Rs[b] + Rc[3 * a + 1] * Rs[3 + b] + Rc[3 * a + 2] * Rs[6 + b];
#pragma unroll
    for (int a = 0; a < 3; a++) pc[a] = sel(act, pn[a], pc[a]);
#pragma unroll
    for (int a = 0; a < 9; a++) Rc[a] = sel(act, Rn[a], Rc[a]);
  }
#pragma unroll
  for (int a = 0; a < 3; a++) { L.z[a] = Rc[3 * a + 2]; L.p[a] = pc[a]; L.pf[a] = quad_bc<3>(pc[a]); }
  wb_body_inertia(tab, c, Rc, pc, L.X);
}

// joint motion vector of my joint in base coordinates: [z ; p x z]; zero on the foot lane
__device__ __forceinline__ void wb_joint_axis(const WbLink &L, int c, double sw[3], double sv[3]) {
  double pz[3];
  cross3(L.p, L.z, pz);
#pragma unroll
  for (int a = 0; a < 3; a++) { sw[a] = sel(c < 3, L.z[a], 0.0); sv[a] = sel(c < 3, pz[a], 0.0); }
}

// Inverse dynamics (recursive Newton-Euler): generalised forces for base velocity V0 = [w ; v], base acceleration
// A0 = [w' ; v' - g_B] (gravity folded in as the usual fictitious acceleration) and my joint's rate / acceleration.
// tau: my joint's generalised force (lanes c < 3); gb[6]: base rows in interface order [force ; moment], replicated.
__device__ __forceinline__ void wb_inverse_dynamics(const WbParamsDev &W, const WbLink &L, int c, const double V0[6],
                                                    const double A0[6], double qd, double qdd, double &tau, double gb[6]) {
  double sw[3], sv[3];
  wb_joint_axis(L, c, sw, sv);
  // velocity of my body: base + the joints up to mine
  double w[3], v[3];
#pragma unroll
  for (int a = 0; a < 3; a++) { w[a] = V0[a] + quad_prefix(sw[a] * qd, c); v[a] = V0[3 + a] + quad_prefix(sv[a] * qd, c); }
  // acceleration: base + sum over the joints up to mine of  S qdd + V x (S qd)
  double jw[3] = {sw[0] * qd, sw[1] * qd, sw[2] * qd}, jv[3] = {sv[0] * qd, sv[1] * qd, sv[2] * qd};
  double c1[3], c2[3], c3[3];
  cross3(w, jw, c1);
  cross3(w, jv, c2);
  cross3(v, jw, c3);
  double aw[3], av[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    aw[a] = A0[a] + quad_prefix(sw[a] * qdd + c1[a], c);
    av[a] = A0[3 + a] + quad_prefix(sv[a] * qdd + c2[a] + c3[a], c);
  }
  // my body's force: X a + V x* (X V)
  double n[3], f[3], nv[3], fv[3];
  inertia_mul(L.X, aw, av, n, f);
  inertia_mul(L.X, w, v, nv, fv);
  double d1[3], d2[3], d3[3];
  cross3(w, nv, d1);
  cross3(v, fv, d2);
  cross3(w, fv, d3);
  double Fc[6];
#pragma unroll
  for (int a = 0; a < 3; a++) { Fc[a] = quad_suffix(n[a] + d1[a] + d2[a], c); Fc[3 + a] = quad_suffix(f[a] + d3[a], c); }
  tau = (sw[0] * Fc[0] + sw[1] * Fc[1] + sw[2] * Fc[2]) + (sv[0] * Fc[3] + sv[1] * Fc[4] + sv[2] * Fc[5]);
  // base rows: the base link's own force + what the four legs transmit
  WbInertia B;
  B.m = W.base_m;
#pragma unroll
  for (int a = 0; a < 3; a++) B.h[a] = W.base_h[a];
#pragma unroll
  for (int a = 0; a < 6; a++) B.I[a] = W.base_I[a];
  double n0[3], f0[3], nv0[3], fv0[3];
  inertia_mul(B, A0, A0 + 3, n0, f0);
  inertia_mul(B, V0, V0 + 3, nv0, fv0);
  cross3(V0, nv0, d1);
  cross3(V0 + 3, fv0, d2);
  cross3(V0, fv0, d3);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    gb[a] = (f0[a] + d3[a]) + legs_root_sum(Fc[3 + a]);
    gb[3 + a] = (n0[a] + d1[a] + d2[a]) + legs_root_sum(Fc[a]);
  }
}

// Composite-rigid-body pass.  total: composite inertia of the whole robot (replicated); Fcol: column of my joint in
// the base block, [moment ; force] = X^c_k S_k; Mleg[j]: entry (my joint, joint j of my leg) of the joint block.
__device__ __forceinline__ void wb_crba(const WbParamsDev &W, const WbLink &L, int c, WbInertia &total, double Fcol[6],
                                        double Mleg[3]) {
  WbInertia Xc;
  Xc.m = quad_suffix(L.X.m, c);
#pragma unroll
  for (int a = 0; a < 3; a++) Xc.h[a] = quad_suffix(L.X.h[a], c);
#pragma unroll
  for (int a = 0; a < 6; a++) Xc.I[a] = quad_suffix(L.X.I[a], c);
  double sw[3], sv[3];
  wb_joint_axis(L, c, sw, sv);
  inertia_mul(Xc, sw, sv, Fcol, Fcol + 3);
  double S[6] = {sw[0], sw[1], sw[2], sv[0], sv[1], sv[2]};
  static_for<3>([&](auto J) {
    constexpr int j = J;
    double dot_own = 0.0, dot_other = 0.0; // S_j . F_mine  and  S_mine . F_j
#pragma unroll
    for (int a = 0; a < 6; a++) {
      dot_own += quad_bc<j>(S[a]) * Fcol[a];
      dot_other += S[a] * quad_bc<j>(Fcol[a]);
    }
    Mleg[j] = sel(j <= c, dot_own, dot_other);
  });
  total.m = W.base_m + legs_root_sum(Xc.m);
#pragma unroll
  for (int a = 0; a < 3; a++) total.h[a] = W.base_h[a] + legs_root_sum(Xc.h[a]);
#pragma unroll
  for (int a = 0; a < 6; a++) total.I[a] = W.base_I[a] + legs_root_sum(Xc.I[a]);
}


// ---- the same dynamics with ONE LANE PER LEG (4 lanes per robot, 16 robots per wavefront): the recursions along a leg
// are serial loops on the lane, only the base terms are sums over the quad.  Nothing is replicated across the lanes of
// a robot, so a wavefront spends about the instructions of the 16-lane form on four times as many robots: the form for
// the dynamics entry, whose 4464 output bytes per robot make it an HBM-bound kernel once the arithmetic is out of the
// way (the whole-body step keeps the 16-lane form: its QP wants the row layout and its batches are latency-bound).
struct WbLeg {
  double z[3][3], p[3][3]; // joint axes and origins, base coordinates
  double pf[3];            // foot-frame origin
  WbInertia X[4];          // the four bodies
};

__device__ __forceinline__ void wb_leg_chain(const CoopTab &tab, const double sj[3], const double cj[3], WbLeg &G) {
  double Rc[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0}, pc[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double R0[9], t[3], Rs[9];
#pragma unroll
    for (int a = 0; a < 9; a++) R0[a] = tab[kTabR0 + 9 * j + a];
#pragma unroll
    for (int a = 0; a < 3; a++) t[a] = tab[kTabXyz + 3 * j + a];
    if (j < 3) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        Rs[a * 3 + 0] = R0[a * 3 + 0] * cj[j] + R0[a * 3 + 1] * sj[j];
        Rs[a * 3 + 1] = R0[a * 3 + 1] * cj[j] - R0[a * 3 + 0] * sj[j];
        Rs[a * 3 + 2] = R0[a * 3 + 2];
      }
    } else {
#pragma unroll
      for (int a = 0; a < 9; a++) Rs[a] = R0[a];
    }
    double Rn[9];
#pragma unroll
    for (int a = 0; a < 3; a++) pc[a] += Rc[3 * a] * t[0] + Rc[3 * a + 1] * t[1] + Rc[3 * a + 2] * t[2];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) Rn[3 * a + b] = Rc[3 * a] * Rs[b] + Rc[3 * a + 1] * Rs[3 + b] + Rc[3 * a + 2] * Rs[6 + b];
#pragma unroll
    for (int a = 0; a < 9; a++) Rc[a] = Rn[a];
    if (j < 3) {
#pragma unroll
      for (int a = 0; a < 3; a++) { G.z[j][a] = Rc[3 * a + 2]; G.p[j][a] = pc[a]; }
    } else {
#pragma unroll
      for (int a = 0; a < 3; a++) G.pf[a] = pc[a];
    }
    wb_body_inertia(tab, j, Rc, pc, G.X[j]);
  }
}

__device__ __forceinline__ void wb_add(WbInertia &A, const WbInertia &Bq) {
  A.m += Bq.m;
#pragma unroll
  for (int a = 0; a < 3; a++) A.h[a] += Bq.h[a];
#pragma unroll
  for (int a = 0; a < 6; a++) A.I[a] += Bq.I[a];
}

// Composite-rigid-body pass of one leg.  Fcol[k]: column of joint k in the base block, [moment ; force];
// Mj[j][k] (j <= k): joint-block entries; total: composite inertia of the whole robot (replicated in the quad).
__device__ __forceinline__ void wb_leg_crba(const WbParamsDev &W, const WbLeg &G, WbInertia &total, double Fcol[3][6],
                                            double Mj[3][3]) {
  WbInertia Xc = G.X[3];
  double S[3][6];
#pragma unroll
  for (int k = 2; k >= 0; k--) {
    wb_add(Xc, G.X[k]);
    double pz[3];
    cross3(G.p[k], G.z[k], pz);
#pragma unroll
    for (int a = 0; a < 3; a++) { S[k][a] = G.z[k][a]; S[k][3 + a] = pz[a]; }
    inertia_mul(Xc, S[k], S[k] + 3, Fcol[k], Fcol[k] + 3);
  }
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int j = 0; j <= k; j++) {
      double d = 0.0;
#pragma unroll
      for (int a = 0; a < 6; a++) d += S[j][a] * Fcol[k][a];
      Mj[j][k] = d;
    }
  total.m = W.base_m + quad_sum(Xc.m);
#pragma unroll
  for (int a = 0; a < 3; a++) total.h[a] = W.base_h[a] + quad_sum(Xc.h[a]);
#pragma unroll
  for (int a = 0; a < 6; a++) total.I[a] = W.base_I[a] + quad_sum(Xc.I[a]);
}

// Recursive Newton-Euler pass of one leg (see wb_inverse_dynamics): tau[k] of the leg's joints, gb[6] the base rows in
// interface order [force ; moment], replicated in the quad.
__device__ __forceinline__ void wb_leg_inverse_dynamics(const WbParamsDev &W, const WbLeg &G, const double V0[6], const double A0[6],
                                                        const double qd[3], const double qdd[3], double tau[3], double gb[6]) {
  double w[3] = {V0[0], V0[1], V0[2]}, v[3] = {V0[3], V0[4], V0[5]};
  double aw[3] = {A0[0], A0[1], A0[2]}, av[3] = {A0[3], A0[4], A0[5]};
  double Fn[4][3], Ff[4][3], S[3][6];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (k < 3) {
      double pz[3];
      cross3(G.p[k], G.z[k], pz);
#pragma unroll
      for (int a = 0; a < 3; a++) { S[k][a] = G.z[k][a]; S[k][3 + a] = pz[a]; }
      double jw[3], jv[3];
#pragma unroll
      for (int a = 0; a < 3; a++) { jw[a] = S[k][a] * qd[k]; jv[a] = S[k][3 + a] * qd[k]; w[a] += jw[a]; v[a] += jv[a]; }
      double c1[3], c2[3], c3[3];
      cross3(w, jw, c1);
      cross3(w, jv, c2);
      cross3(v, jw, c3);
#pragma unroll
      for (int a = 0; a < 3; a++) { aw[a] += S[k][a] * qdd[k] + c1[a]; av[a] += S[k][3 + a] * qdd[k] + c2[a] + c3[a]; }
    }
    double n[3], f[3], nv[3], fv[3], d1[3], d2[3], d3[3];
    inertia_mul(G.X[k], aw, av, n, f);
    inertia_mul(G.X[k], w, v, nv, fv);
    cross3(w, nv, d1);
    cross3(v, fv, d2);
    cross3(w, fv, d3);
#pragma unroll
    for (int a = 0; a < 3; a++) { Fn[k][a] = n[a] + d1[a] + d2[a]; Ff[k][a] = f[a] + d3[a]; }
  }
#pragma unroll
  for (int k = 2; k >= 0; k--) {
#pragma unroll
    for (int a = 0; a < 3; a++) { Fn[k][a] += Fn[k + 1][a]; Ff[k][a] += Ff[k + 1][a]; }
    tau[k] = (S[k][0] * Fn[k][0] + S[k][1] * Fn[k][1] + S[k][2] * Fn[k][2]) +
             (S[k][3] * Ff[k][0] + S[k][4] * Ff[k][1] + S[k][5] * Ff[k][2]);
  }
  WbInertia Bq;
  Bq.m = W.base_m;
#pragma unroll
  for (int a = 0; a < 3; a++) Bq.h[a] = W.base_h[a];
#pragma unroll
  for (int a = 0; a < 6; a++) Bq.I[a] = W.base_I[a];
  double n0[3], f0[3], nv0[3], fv0[3], d1[3], d2[3], d3[3];
  inertia_mul(Bq, A0, A0 + 3, n0, f0);
  inertia_mul(Bq, V0, V0 + 3, nv0, fv0);
  cross3(V0, nv0, d1);
  cross3(V0 + 3, fv0, d2);
  cross3(V0, fv0, d3);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    gb[a] = (f0[a] + d3[a]) + quad_sum(Ff[0][a]);
    gb[3 + a] = (n0[a] + d1[a] + d2[a]) + quad_sum(Fn[0][a]);
  }
}

// Layout of one robot's staging block in LDS for the dynamics kernel: M [18][18]; then h [18], Jc [12][18]
// (the kernel stages M first and, after writing it out, h and Jc in the same block)
constexpr int kWbM = 0, kWbStage = 324; // pass 2: h of the block's 4 robots [4][18], then Jc [4][216]

} // namespace coop
} // namespace qlamd
