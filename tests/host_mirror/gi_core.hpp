// Dense Goldfarb-Idnani dual active-set QP for one problem per lane (host/device).
//
//   min 1/2 x'Gx + g0'x   s.t.  CE'x + ce0 = 0,  CI'x + ci0 >= 0
//
// Replaces quadprogpp::solve_quadprog (qp_solver/src/QuadProg++.cc:52-446) behind
// qp_solver::QuadraticProblemSolver::minimize (qp_solver/src/quadraticproblemsolver.cpp:65-97).
// The iteration -- Cholesky, J = L^-T, most-violated constraint, partial / full steps, Givens
// add / drop, termination test -- follows the reference step by step on purpose: the reference's
// pose optimisation always passes one all-zero equality column, which the solver "adds" with a
// zero pivot and then keeps every later step inside span(J[:,1:]).  Reproducing
// those answers needs the same J, not just the same minimiser.
//
// All arrays live behind a `Scr` accessor (LDS or global, [element][problem] layout) because
// every index is data dependent.  Sizes are template maxima; n, p, m are run-time.
#pragma once

#include <math.h>
#include <stdint.h>

#include "balance_core.hpp" // QL_HD, status codes

namespace qlamd {

template <int NMAX, int PMAX, int MMAX>
struct GiLayout {
  static constexpr int kQ = MMAX + PMAX + 1;
  static constexpr int G = 0;                    // n x n, becomes L (mirrored to the upper triangle)
  static constexpr int J = G + NMAX * NMAX;
  static constexpr int R = J + NMAX * NMAX;
  static constexpr int CI = R + NMAX * NMAX;     // n x m
  static constexpr int CE = CI + NMAX * MMAX;    // n x p
  static constexpr int G0 = CE + NMAX * PMAX;
  static constexpr int CI0 = G0 + NMAX;
  static constexpr int CE0 = CI0 + MMAX;
  static constexpr int X = CE0 + PMAX;
  static constexpr int D = X + NMAX;
  static constexpr int Z = D + NMAX;
  static constexpr int NP = Z + NMAX;
  static constexpr int XOLD = NP + NMAX;
  static constexpr int S = XOLD + NMAX;
  static constexpr int RR = S + kQ;
  static constexpr int U = RR + kQ;
  static constexpr int UOLD = U + kQ;
  static constexpr int A = UOLD + kQ;            // working set (ints stored as doubles)
  static constexpr int AOLD = A + kQ;
  static constexpr int kTotal = AOLD + kQ;
};

// hypot without overflow, QuadProg++.cc:647-664
QL_HD double gi_hyp(double a, double b) {
  const double a1 = fabs(a), b1 = fabs(b);
  if (a1 > b1) { const double t = b1 / a1; return a1 * sqrt(1.0 + t * t); }
  if (b1 > a1) { const double t = a1 / b1; return b1 * sqrt(1.0 + t * t); }
  return a1 * sqrt(2.0);
}

// The caller fills G, G0, CE, CE0, CI, CI0 in `s`; x comes back in s[X..X+n).  Returns a status
// (kStatusOk / kStatusInfeasible / kStatusNotPd / kStatusMaxIter); *f_out the objective value.
template <int NMAX, int PMAX, int MMAX, class Scr>
QL_HD int gi_solve(Scr &s, int n, int p, int m, double *f_out, int *iters_out) {
  typedef GiLayout<NMAX, PMAX, MMAX> Ly;
  const double eps = 2.220446049250313e-16, inf = INFINITY;
#define GI_M(base, i, j) s.at((base) + (i) * n + (j))
  const auto dot = [&](int a, int b) { double acc = 0.0; for (int i = 0; i < n; i++) acc += s.at(a + i) * s.at(b + i); return acc; };

  // ---- preprocessing, QuadProg++.cc:117-167
  double c1 = 0.0, c2 = 0.0;
  for (int i = 0; i < n; i++) c1 += GI_M(Ly::G, i, i);
  for (int i = 0; i < n; i++) { // in-place Cholesky, :678-709
    for (int j = i; j < n; j++) {
      double acc = GI_M(Ly::G, i, j);
      for (int k = i - 1; k >= 0; k--) acc -= GI_M(Ly::G, i, k) * GI_M(Ly::G, j, k);
      if (i == j) {
        if (!(acc > 0.0)) { *f_out = NAN; return kStatusNotPd; }
        GI_M(Ly::G, i, i) = sqrt(acc);
      } else {
        GI_M(Ly::G, j, i) = acc / GI_M(Ly::G, i, i);
      }
    }
    for (int k = i + 1; k < n; k++) GI_M(Ly::G, i, k) = GI_M(Ly::G, k, i);
  }
  const auto forward = [&](int y, int b) { // :722-734
    for (int i = 0; i < n; i++) {
      double acc = s.at(b + i);
      for (int j = 0; j < i; j++) acc -= GI_M(Ly::G, i, j) * s.at(y + j);
      s.at(y + i) = acc / GI_M(Ly::G, i, i);
    }
  };
  for (int i = 0; i < n * n; i++) s.at(Ly::R + i) = 0.0;
  for (int i = 0; i < n; i++) s.at(Ly::D + i) = 0.0;
  double rnorm = 1.0;
  for (int i = 0; i < n; i++) { // J = L^-T, :139-147
    s.at(Ly::D + i) = 1.0;
    forward(Ly::Z, Ly::D);
    for (int j = 0; j < n; j++) GI_M(Ly::J, i, j) = s.at(Ly::Z + j);
    c2 += s.at(Ly::Z + i);
    s.at(Ly::D + i) = 0.0;
  }
  forward(Ly::Z, Ly::G0); // x = -G^-1 g0, :159-161
  for (int i = n - 1; i >= 0; i--) {
    double acc = s.at(Ly::Z + i);
    for (int j = i + 1; j < n; j++) acc -= GI_M(Ly::G, i, j) * s.at(Ly::X + j);
    s.at(Ly::X + i) = acc / GI_M(Ly::G, i, i);
  }
  for (int i = 0; i < n; i++) s.at(Ly::X + i) = -s.at(Ly::X + i);
  double f_value = 0.5 * dot(Ly::G0, Ly::X);

  int iq = 0;
  const auto compute_d = [&]() { // d = J' np, :448-461
    for (int i = 0; i < n; i++) {
      double acc = 0.0;
      for (int j = 0; j < n; j++) acc += GI_M(Ly::J, j, i) * s.at(Ly::NP + j);
      s.at(Ly::D + i) = acc;
    }
  };
  const auto update_z = [&]() { // z = J[:, iq:] d[iq:], :463-474
    for (int i = 0; i < n; i++) {
      double acc = 0.0;
      for (int j = iq; j < n; j++) acc += GI_M(Ly::J, i, j) * s.at(Ly::D + j);
      s.at(Ly::Z + i) = acc;
    }
  };
  const auto update_r = [&]() { // r = R^-1 d, :476-489
    for (int i = iq - 1; i >= 0; i--) {
      double acc = 0.0;
      for (int j = i + 1; j < iq; j++) acc += GI_M(Ly::R, i, j) * s.at(Ly::RR + j);
      s.at(Ly::RR + i) = (s.at(Ly::D + i) - acc) / GI_M(Ly::R, i, i);
    }
  };
  const auto add_constraint = [&]() -> bool { // :491-560
    for (int j = n - 1; j >= iq + 1; j--) {
      double cc = s.at(Ly::D + j - 1), ss = s.at(Ly::D + j);
      const double h = gi_hyp(cc, ss);
      if (fabs(h) < eps) continue;
      s.at(Ly::D + j) = 0.0;
      ss = ss / h; cc = cc / h;
      if (cc < 0.0) { cc = -cc; ss = -ss; s.at(Ly::D + j - 1) = -h; }
      else s.at(Ly::D + j - 1) = h;
      const double xny = ss / (1.0 + cc);
      for (int k = 0; k < n; k++) {
        const double t1 = GI_M(Ly::J, k, j - 1), t2 = GI_M(Ly::J, k, j);
        const double nv = t1 * cc + t2 * ss;
        GI_M(Ly::J, k, j - 1) = nv;
        GI_M(Ly::J, k, j) = xny * (t1 + nv) - t2;
      }
    }
    iq++;
    for (int i = 0; i < iq; i++) GI_M(Ly::R, i, iq - 1) = s.at(Ly::D + i);
    const double dq = fabs(s.at(Ly::D + iq - 1));
    if (dq <= eps * rnorm) return false;
    rnorm = fmax(rnorm, dq);
    return true;
  };
  const auto delete_constraint = [&](int l) -> bool { // :562-645
    int qq = -1;
    for (int i = p; i < iq; i++)
      if ((int)s.at(Ly::A + i) == l) { qq = i; break; }
    if (qq < 0) return false;
    for (int i = qq; i < iq - 1; i++) {
      s.at(Ly::A + i) = s.at(Ly::A + i + 1);
      s.at(Ly::U + i) = s.at(Ly::U + i + 1);
      for (int j = 0; j < n; j++) GI_M(Ly::R, j, i) = GI_M(Ly::R, j, i + 1);
    }
    s.at(Ly::A + iq - 1) = s.at(Ly::A + iq);
    s.at(Ly::U + iq - 1) = s.at(Ly::U + iq);
    s.at(Ly::A + iq) = 0.0;
    s.at(Ly::U + iq) = 0.0;
    for (int j = 0; j < iq; j++) GI_M(Ly::R, j, iq - 1) = 0.0;
    iq--;
    if (iq == 0) return true;
    for (int j = qq; j < iq; j++) {
      double cc = GI_M(Ly::R, j, j), ss = GI_M(Ly::R, j + 1, j);
      const double h = gi_hyp(cc, ss);
      if (fabs(h) < eps) continue;
      cc = cc / h; ss = ss / h;
      GI_M(Ly::R, j + 1, j) = 0.0;
      if (cc < 0.0) { GI_M(Ly::R, j, j) = -h; cc = -cc; ss = -ss; }
      else GI_M(Ly::R, j, j) = h;
      const double xny = ss / (1.0 + cc);
      for (int k = j + 1; k < iq; k++) {
        const double t1 = GI_M(Ly::R, j, k), t2 = GI_M(Ly::R, j + 1, k);
        const double nv = t1 * cc + t2 * ss;
        GI_M(Ly::R, j, k) = nv;
        GI_M(Ly::R, j + 1, k) = xny * (t1 + nv) - t2;
      }
      for (int k = 0; k < n; k++) {
        const double t1 = GI_M(Ly::J, k, j), t2 = GI_M(Ly::J, k, j + 1);
        const double nv = t1 * cc + t2 * ss;
        GI_M(Ly::J, k, j) = nv;
        GI_M(Ly::J, k, j + 1) = xny * (nv + t1) - t2;
      }
    }
    return true;
  };

  // ---- equality constraints, :169-210 (a failed add is ignored there, which is what makes the
  // reference's dummy zero column harmful)
  for (int i = 0; i < p; i++) {
    for (int j = 0; j < n; j++) s.at(Ly::NP + j) = s.at(Ly::CE + j * p + i);
    compute_d(); update_z(); update_r();
    double t2 = 0.0;
    if (fabs(dot(Ly::Z, Ly::Z)) > eps) t2 = (-dot(Ly::NP, Ly::X) - s.at(Ly::CE0 + i)) / dot(Ly::Z, Ly::NP);
    for (int k = 0; k < n; k++) s.at(Ly::X + k) += t2 * s.at(Ly::Z + k);
    s.at(Ly::U + iq) = t2;
    for (int k = 0; k < iq; k++) s.at(Ly::U + k) -= t2 * s.at(Ly::RR + k);
    f_value += 0.5 * (t2 * t2) * dot(Ly::Z, Ly::NP);
    s.at(Ly::A + i) = (double)(-i - 1);
    (void)add_constraint();
  }

  uint64_t active = 0, allowed = ~0ull; // iai[i] == -1  <=>  bit i of `active`; iaexcl <=> `allowed`
  int status = kStatusOk, iter = 0, ip = 0, l = 0;
  double ss = 0.0;
  enum { L1, L2, L2A } at = L1;
  bool finished = false;
  for (int guard = 0; guard < 4000 && !finished; guard++) {
    if (at == L1) { // :216-260
      iter++;
      for (int i = p; i < iq; i++) active |= 1ull << (int)s.at(Ly::A + i);
      ss = 0.0; ip = 0;
      double psi = 0.0;
      allowed = ~0ull;
      for (int i = 0; i < m; i++) {
        double acc = 0.0;
        for (int j = 0; j < n; j++) acc += s.at(Ly::CI + j * m + i) * s.at(Ly::X + j);
        acc += s.at(Ly::CI0 + i);
        s.at(Ly::S + i) = acc;
        psi += fmin(0.0, acc);
      }
      if (fabs(psi) <= m * eps * c1 * c2 * 100.0) { finished = true; break; }
      for (int i = 0; i < iq; i++) { s.at(Ly::UOLD + i) = s.at(Ly::U + i); s.at(Ly::AOLD + i) = s.at(Ly::A + i); }
      for (int i = 0; i < n; i++) s.at(Ly::XOLD + i) = s.at(Ly::X + i);
      at = L2;
    }
    if (at == L2) { // :262-287
      for (int i = 0; i < m; i++)
        if (s.at(Ly::S + i) < ss && !((active >> i) & 1ull) && ((allowed >> i) & 1ull)) { ss = s.at(Ly::S + i); ip = i; }
      if (ss >= 0.0) { finished = true; break; }
      for (int i = 0; i < n; i++) s.at(Ly::NP + i) = s.at(Ly::CI + i * m + ip);
      s.at(Ly::U + iq) = 0.0;
      s.at(Ly::A + iq) = (double)ip;
      at = L2A;
    }
    // l2a, :289-331
    compute_d(); update_z(); update_r();
    l = 0;
    double t1 = inf;
    for (int k = p; k < iq; k++) {
      const double rk = s.at(Ly::RR + k);
      if (rk > 0.0 && s.at(Ly::U + k) / rk < t1) { t1 = s.at(Ly::U + k) / rk; l = (int)s.at(Ly::A + k); }
    }
    double t2;
    if (fabs(dot(Ly::Z, Ly::Z)) > eps) {
      t2 = -s.at(Ly::S + ip) / dot(Ly::Z, Ly::NP);
      if (t2 < 0) t2 = inf;
    } else {
      t2 = inf;
    }
    const double t = fmin(t1, t2);
    if (t >= inf) { status = kStatusInfeasible; f_value = inf; finished = true; break; } // :339-344
    if (t2 >= inf) { // dual step, :346-362
      for (int k = 0; k < iq; k++) s.at(Ly::U + k) -= t * s.at(Ly::RR + k);
      s.at(Ly::U + iq) += t;
      active &= ~(1ull << l);
      if (!delete_constraint(l)) { status = kStatusMaxIter; finished = true; break; }
      at = L2A;
      continue;
    }
    for (int k = 0; k < n; k++) s.at(Ly::X + k) += t * s.at(Ly::Z + k); // :364-374
    f_value += t * dot(Ly::Z, Ly::NP) * (0.5 * t + s.at(Ly::U + iq));
    for (int k = 0; k < iq; k++) s.at(Ly::U + k) -= t * s.at(Ly::RR + k);
    s.at(Ly::U + iq) += t;
    if (fabs(t - t2) < eps) { // full step, :384-421
      if (!add_constraint()) {
        allowed &= ~(1ull << ip);
        if (!delete_constraint(ip)) { status = kStatusMaxIter; finished = true; break; }
        active = 0;
        for (int i = p; i < iq; i++) {
          s.at(Ly::A + i) = s.at(Ly::AOLD + i);
          s.at(Ly::U + i) = s.at(Ly::UOLD + i);
          active |= 1ull << (int)s.at(Ly::A + i);
        }
        for (int i = 0; i < n; i++) s.at(Ly::X + i) = s.at(Ly::XOLD + i);
        at = L2;
        continue;
      }
      active |= 1ull << ip;
      at = L1;
      continue;
    }
    active &= ~(1ull << l); // partial step, :423-445
    if (!delete_constraint(l)) { status = kStatusMaxIter; finished = true; break; }
    {
      double acc = 0.0;
      for (int k = 0; k < n; k++) acc += s.at(Ly::CI + k * m + ip) * s.at(Ly::X + k);
      s.at(Ly::S + ip) = acc + s.at(Ly::CI0 + ip);
    }
    at = L2A;
  }
#undef GI_M
  if (!finished && status == kStatusOk) status = kStatusMaxIter;
  *f_out = f_value;
  if (iters_out) *iters_out = iter;
  return status;
}

} // namespace qlamd
