// Goldfarb-Idnani for the pose-optimisation inner QP (n = 6, p <= 1, m <= 8), register resident.
//
// Same algorithm, same order of floating-point operations as gi_core.hpp (which follows
// qp_solver/src/QuadProg++.cc:52-748 step by step) -- the host build of both is bit-identical to
// the oracle -- but with L, J, R and the small vectors in registers: every loop is fully unrolled
// over the compile-time maxima and the run-time extents (iq, positions in the working set) become
// predicates.  Only the constraint matrix CI stays behind the `Scr` accessor (LDS), because one
// COLUMN of it is picked by a run-time index.  The LDS-resident gi_core version spends most of its
// time waiting on dependent LDS reads (97 us per 4096 pose solves); this one does not.
#pragma once

#include "gi_core.hpp"

namespace qlamd {

struct Gi6Layout { // scratch elements used by gi6_solve: CI (6 x 8, row-major with stride m) and ci0
  static constexpr int CI = 0, CI0 = 48, kTotal = 56;
};

// G (6x6 row-major, destroyed), g0, CE (6 x p), ce0 in registers; CI / ci0 behind `s`.
template <class Scr>
QL_HD int gi6_solve(Scr &s, double (&G)[36], const double (&g0)[6], const double (&CE)[6], double ce0, int p, int m,
                    double (&x)[6], double *f_out) {
  constexpr int N = 6, Q = 10; // Q >= m + p + 1
  const double eps = 2.220446049250313e-16, inf = INFINITY;
  double J[36], R[36], d[N], z[N], np[N], xold[N];
  double sv[Q], r[Q], u[Q], uold[Q];
  int A[Q], Aold[Q];
  QL_UNROLL for (int i = 0; i < Q; i++) { sv[i] = 0.0; r[i] = 0.0; u[i] = 0.0; uold[i] = 0.0; A[i] = 0; Aold[i] = 0; }
  QL_UNROLL for (int i = 0; i < 36; i++) { R[i] = 0.0; J[i] = 0.0; }
  QL_UNROLL for (int i = 0; i < N; i++) { d[i] = 0.0; z[i] = 0.0; np[i] = 0.0; xold[i] = 0.0; x[i] = 0.0; }

  // ---- preprocessing, QuadProg++.cc:117-167
  double c1 = 0.0, c2 = 0.0;
  QL_UNROLL for (int i = 0; i < N; i++) c1 += G[i * N + i];
  bool not_pd = false;
  QL_UNROLL for (int i = 0; i < N; i++) {
    QL_UNROLL for (int j = i; j < N; j++) {
      double acc = G[i * N + j];
      QL_UNROLL for (int k = i - 1; k >= 0; k--) acc -= G[i * N + k] * G[j * N + k];
      if (i == j) {
        if (!(acc > 0.0)) not_pd = true;
        G[i * N + i] = sqrt(acc);
      } else {
        G[j * N + i] = acc / G[i * N + i];
      }
    }
    QL_UNROLL for (int k = i + 1; k < N; k++) G[i * N + k] = G[k * N + i];
  }
  if (not_pd) { *f_out = NAN; return kStatusNotPd; }
  const auto forward = [&](double (&y)[N], const double (&b)[N]) {
    QL_UNROLL for (int i = 0; i < N; i++) {
      double acc = b[i];
      QL_UNROLL for (int j = 0; j < i; j++) acc -= G[i * N + j] * y[j];
      y[i] = acc / G[i * N + i];
    }
  };
  double rnorm = 1.0;
  QL_UNROLL for (int i = 0; i < N; i++) {
    d[i] = 1.0;
    forward(z, d);
    QL_UNROLL for (int j = 0; j < N; j++) J[i * N + j] = z[j];
    c2 += z[i];
    d[i] = 0.0;
  }
  forward(z, g0);
  QL_UNROLL for (int i = N - 1; i >= 0; i--) {
    double acc = z[i];
    QL_UNROLL for (int j = i + 1; j < N; j++) acc -= G[i * N + j] * x[j];
    x[i] = acc / G[i * N + i];
  }
  QL_UNROLL for (int i = 0; i < N; i++) x[i] = -x[i];
  const auto dot6 = [&](const double (&a)[N], const double (&b)[N]) {
    double acc = 0.0;
    QL_UNROLL for (int i = 0; i < N; i++) acc += a[i] * b[i];
    return acc;
  };
  double f_value = 0.5 * dot6(g0, x);

  int iq = 0;
  const auto compute_d = [&]() {
    QL_UNROLL for (int i = 0; i < N; i++) {
      double acc = 0.0;
      QL_UNROLL for (int j = 0; j < N; j++) acc += J[j * N + i] * np[j];
      d[i] = acc;
    }
  };
  const auto update_z = [&]() {
    QL_UNROLL for (int i = 0; i < N; i++) {
      double acc = 0.0;
      QL_UNROLL for (int j = 0; j < N; j++)
        if (j >= iq) acc += J[i * N + j] * d[j];
      z[i] = acc;
    }
  };
  const auto update_r = [&]() {
    QL_UNROLL for (int i = N - 1; i >= 0; i--) {
      if (i < iq) {
        double acc = 0.0;
        QL_UNROLL for (int j = i + 1; j < N; j++)
          if (j < iq) acc += R[i * N + j] * r[j];
        r[i] = (d[i] - acc) / R[i * N + i];
      }
    }
  };
  const auto add_constraint = [&]() -> bool {
    QL_UNROLL for (int j = N - 1; j >= 1; j--) {
      if (j >= iq + 1) {
        double cc = d[j - 1], ss = d[j];
        const double h = gi_hyp(cc, ss);
        if (!(fabs(h) < eps)) {
          d[j] = 0.0;
          ss = ss / h; cc = cc / h;
          if (cc < 0.0) { cc = -cc; ss = -ss; d[j - 1] = -h; }
          else d[j - 1] = h;
          const double xny = ss / (1.0 + cc);
          QL_UNROLL for (int k = 0; k < N; k++) {
            const double t1 = J[k * N + j - 1], t2 = J[k * N + j];
            const double nv = t1 * cc + t2 * ss;
            J[k * N + j - 1] = nv;
            J[k * N + j] = xny * (t1 + nv) - t2;
          }
        }
      }
    }
    iq++;
    double dq = 0.0;
    QL_UNROLL for (int c = 0; c < N; c++) {
      if (c == iq - 1) {
        QL_UNROLL for (int i = 0; i <= c; i++) R[i * N + c] = d[i];
        dq = fabs(d[c]);
      }
    }
    if (dq <= eps * rnorm) return false;
    rnorm = fmax(rnorm, dq);
    return true;
  };
  const auto delete_constraint = [&](int l) -> bool {
    int qq = -1;
    QL_UNROLL for (int i = Q - 1; i >= 0; i--)
      if (i >= p && i < iq && A[i] == l) qq = i; // lowest matching position
    if (qq < 0) return false;
    QL_UNROLL for (int i = 0; i < Q - 1; i++) {
      if (i >= qq && i < iq - 1) {
        A[i] = A[i + 1];
        u[i] = u[i + 1];
        if (i < N - 1) {
          QL_UNROLL for (int j = 0; j < N; j++) R[j * N + i] = R[j * N + (i + 1 < N ? i + 1 : N - 1)];
        }
      }
    }
    // A[iq-1] = A[iq]; u[iq-1] = u[iq]; A[iq] = 0; u[iq] = 0; R[:, iq-1] = 0   (:595-600)
    QL_UNROLL for (int i = 0; i < Q - 1; i++) {
      if (i == iq - 1) { A[i] = A[i + 1]; u[i] = u[i + 1]; }
    }
    QL_UNROLL for (int i = 0; i < Q; i++) {
      if (i == iq) { A[i] = 0; u[i] = 0.0; }
    }
    QL_UNROLL for (int c = 0; c < N; c++) {
      if (c == iq - 1) {
        QL_UNROLL for (int j = 0; j < N; j++)
          if (j < iq) R[j * N + c] = 0.0;
      }
    }
    iq--;
    if (iq == 0) return true;
    QL_UNROLL for (int j = 0; j < N - 1; j++) {
      if (j >= qq && j < iq) {
        double cc = R[j * N + j], ss = R[(j + 1) * N + j];
        const double h = gi_hyp(cc, ss);
        if (!(fabs(h) < eps)) {
          cc = cc / h; ss = ss / h;
          R[(j + 1) * N + j] = 0.0;
          if (cc < 0.0) { R[j * N + j] = -h; cc = -cc; ss = -ss; }
          else R[j * N + j] = h;
          const double xny = ss / (1.0 + cc);
          QL_UNROLL for (int k = j + 1; k < N; k++) {
            if (k < iq) {
              const double t1 = R[j * N + k], t2 = R[(j + 1) * N + k];
              const double nv = t1 * cc + t2 * ss;
              R[j * N + k] = nv;
              R[(j + 1) * N + k] = xny * (t1 + nv) - t2;
            }
          }
          QL_UNROLL for (int k = 0; k < N; k++) {
            const double t1 = J[k * N + j], t2 = J[k * N + j + 1];
            const double nv = t1 * cc + t2 * ss;
            J[k * N + j] = nv;
            J[k * N + j + 1] = xny * (nv + t1) - t2;
          }
        }
      }
    }
    return true;
  };
  // dynamic-position accessors on the small register vectors
  const auto set_at = [&](double (&v)[Q], int pos, double val) {
    QL_UNROLL for (int i = 0; i < Q; i++)
      if (i == pos) v[i] = val;
  };
  const auto add_at = [&](double (&v)[Q], int pos, double val) {
    QL_UNROLL for (int i = 0; i < Q; i++)
      if (i == pos) v[i] += val;
  };
  const auto get_at = [&](const double (&v)[Q], int pos) -> double {
    double o = 0.0;
    QL_UNROLL for (int i = 0; i < Q; i++) o = (i == pos) ? v[i] : o;
    return o;
  };

  // ---- equality constraint (at most one), :169-210
  if (p > 0) {
    QL_UNROLL for (int j = 0; j < N; j++) np[j] = CE[j];
    compute_d(); update_z(); update_r();
    double t2 = 0.0;
    if (fabs(dot6(z, z)) > eps) t2 = (-dot6(np, x) - ce0) / dot6(z, np);
    QL_UNROLL for (int k = 0; k < N; k++) x[k] += t2 * z[k];
    set_at(u, iq, t2);
    QL_UNROLL for (int k = 0; k < Q; k++)
      if (k < iq) u[k] -= t2 * r[k];
    f_value += 0.5 * (t2 * t2) * dot6(z, np);
    A[0] = -1;
    (void)add_constraint();
  }

  unsigned active = 0, allowed = ~0u;
  int status = kStatusOk, ip = 0, l = 0;
  double ss = 0.0;
  enum { L1, L2, L2A } at = L1;
  bool finished = false;
  for (int guard = 0; guard < 4000 && !finished; guard++) {
    if (at == L1) {
      QL_UNROLL for (int i = 0; i < Q; i++)
        if (i >= p && i < iq) active |= 1u << A[i];
      ss = 0.0; ip = 0;
      double psi = 0.0;
      allowed = ~0u;
      QL_UNROLL for (int i = 0; i < 8; i++) {
        if (i < m) {
          double acc = 0.0;
          QL_UNROLL for (int j = 0; j < N; j++) acc += s.at(Gi6Layout::CI + j * m + i) * x[j];
          acc += s.at(Gi6Layout::CI0 + i);
          sv[i] = acc;
          psi += fmin(0.0, acc);
        }
      }
      if (fabs(psi) <= m * eps * c1 * c2 * 100.0) { finished = true; break; }
      QL_UNROLL for (int i = 0; i < Q; i++)
        if (i < iq) { uold[i] = u[i]; Aold[i] = A[i]; }
      QL_UNROLL for (int i = 0; i < N; i++) xold[i] = x[i];
      at = L2;
    }
    if (at == L2) {
      QL_UNROLL for (int i = 0; i < 8; i++)
        if (i < m && sv[i] < ss && !((active >> i) & 1u) && ((allowed >> i) & 1u)) { ss = sv[i]; ip = i; }
      if (ss >= 0.0) { finished = true; break; }
      QL_UNROLL for (int i = 0; i < N; i++) np[i] = s.at(Gi6Layout::CI + i * m + ip);
      set_at(u, iq, 0.0);
      QL_UNROLL for (int i = 0; i < Q; i++)
        if (i == iq) A[i] = ip;
      at = L2A;
    }
    compute_d(); update_z(); update_r();
    l = 0;
    double t1 = inf;
    QL_UNROLL for (int k = 0; k < Q; k++) {
      if (k >= p && k < iq && r[k] > 0.0 && u[k] / r[k] < t1) { t1 = u[k] / r[k]; l = A[k]; }
    }
    const double sip = get_at(sv, ip);
    double t2;
    if (fabs(dot6(z, z)) > eps) {
      t2 = -sip / dot6(z, np);
      if (t2 < 0) t2 = inf;
    } else {
      t2 = inf;
    }
    const double t = fmin(t1, t2);
    if (t >= inf) { status = kStatusInfeasible; f_value = inf; finished = true; break; }
    if (t2 >= inf) {
      QL_UNROLL for (int k = 0; k < Q; k++)
        if (k < iq) u[k] -= t * r[k];
      add_at(u, iq, t);
      active &= ~(1u << l);
      if (!delete_constraint(l)) { status = kStatusMaxIter; finished = true; break; }
      at = L2A;
      continue;
    }
    QL_UNROLL for (int k = 0; k < N; k++) x[k] += t * z[k];
    f_value += t * dot6(z, np) * (0.5 * t + get_at(u, iq));
    QL_UNROLL for (int k = 0; k < Q; k++)
      if (k < iq) u[k] -= t * r[k];
    add_at(u, iq, t);
    if (fabs(t - t2) < eps) {
      if (!add_constraint()) {
        allowed &= ~(1u << ip);
        if (!delete_constraint(ip)) { status = kStatusMaxIter; finished = true; break; }
        active = 0;
        QL_UNROLL for (int i = 0; i < Q; i++) {
          if (i >= p && i < iq) {
            A[i] = Aold[i];
            u[i] = uold[i];
            active |= 1u << A[i];
          }
        }
        QL_UNROLL for (int i = 0; i < N; i++) x[i] = xold[i];
        at = L2;
        continue;
      }
      active |= 1u << ip;
      at = L1;
      continue;
    }
    active &= ~(1u << l);
    if (!delete_constraint(l)) { status = kStatusMaxIter; finished = true; break; }
    {
      double acc = 0.0;
      QL_UNROLL for (int k = 0; k < N; k++) acc += s.at(Gi6Layout::CI + k * m + ip) * x[k];
      set_at(sv, ip, acc + s.at(Gi6Layout::CI0 + ip));
    }
    at = L2A;
  }
  if (!finished && status == kStatusOk) status = kStatusMaxIter;
  *f_out = f_value;
  return status;
}

} // namespace qlamd
