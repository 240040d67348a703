// Lane-cooperative dense QP batch: 16 lanes per problem, 4 problems per wavefront.  Device-only.
//
// Replaces quadprogpp::solve_quadprog / qp_solver::QuadraticProblemSolver::minimize
// (qp_solver/src/QuadProg++.cc:52-446, qp_solver/src/quadraticproblemsolver.cpp:65-97) for
//   min 1/2 x'Gx + g0'x   s.t.  CE'x + ce0 = 0,  CI'x + ci0 >= 0,     n <= N, at most two equalities,
//   m <= 24 with two inequalities per lane (KC = 2), m <= 48 with three (KC = 3, the whole-body QP),
// with the Goldfarb-Idnani method in the explicit-operator form of balance_coop.hpp / pose_coop.hpp:
//   lane i < n      variable lane: component i of x, g0, z; row i of G and of the projector H
//   lane k < n      slot lane: row k of N*, multiplier and constraint id of active-set slot k
//   lane j          constraint lane: inequalities j, j + 16 (and j + 32 for KC = 3): normal, ci0, slack
// Rows n..N-1 are padded with the identity (their variables stay 0).  Pivot rule, step lengths and the
// termination test are QuadProg++'s; exact ties between equally violated constraints go to the lowest lane.
//
// The single equality: a genuine one is stepped onto and projected out (H -= z z'/z'n) before the inequality
// loop, as QuadProg++ does (:169-210).  An all-zero column -- what the reference's wrapper always passes
//  -- consumes the first column of J = L^-T there without moving x, i.e. it is the equality
// (G e1)'(x - x0) = 0: H0 = G^-1 - e1 e1'/G11 (tests/test_oracle_quadprog.py).  A second equality column is
// projected out the same way after the first; it has to be a genuine normal (an all-zero or linearly dependent
// second column is ignored, where the reference would spend another column of J on it: no caller in the reference
// passes two columns).
#pragma once

#include "balance_coop.hpp"

namespace qlamd {
namespace coop {

constexpr int kQpCoopRows = 4;
template <int N, int KC = 2>
struct QpCoopLds {
  enum { kRows = KC == 2 ? 24 : 16 * KC, kCt = 0, kNrow = kRows * N, kNst = kRows * N + N, kTotal = kRows * N + N + N * N };
};

// Gm: row lr of G (identity row for a variable that is not free, zero for lr >= N); g0: component lr; ne: component
// lr of the equality normal (has_eq), ce0 its offset; a[s]/b[s]/v[s]: normal, offset and validity of inequality
// lr + 16 s.  n_free: dimension of the subspace the constraints act on (= n unless some rows of G are identity
// padding in the middle, as for the swing legs of the whole-body QP); it only enters the empty-null-space rule.
// Returns the status; x_out = component lr of the minimiser, f_out = objective value (replicated).
template <int N, int KC>
__device__ __forceinline__ int qp_coop_impl(const double (&Gm)[N], double g0, int n, int n_free, int m, bool has_eq, double ne,
                                            double ce0, const double (&a)[KC][N], const double (&b)[KC], const bool (&v)[KC],
                                            bool skip, double *lds_row, double &x_out, double &f_out, bool has_eq2 = false,
                                            double ne2 = 0.0, double ce02 = 0.0) {
  typedef QpCoopLds<N, KC> L;
  typedef typename std::conditional<(KC > 2), unsigned long long, unsigned>::type mask_t;
  const int lr = threadIdx.x & 15;
  const bool var = lr < n;
  const double eps = 2.220446049250313e-16;
  const double inf = INFINITY;
  double *ct = lds_row + L::kCt, *nrow = lds_row + L::kNrow, *nst = lds_row + L::kNst;

  // normals by constraint into LDS: variable lane i reads a_p[i] = ct[N p + i]
#pragma unroll
  for (int s = 0; s < KC; s++)
    if (lr + 16 * s < m) {
#pragma unroll
      for (int i = 0; i < N; i++) ct[N * (lr + 16 * s) + i] = a[s][i];
    }
  double diag = 0.0;
#pragma unroll
  for (int j = 0; j < N; j++) diag = sel(lr == j, Gm[j], diag);
  const double c1 = row_sum(sel(var, diag, 0.0));
  double H[N];
#pragma unroll
  for (int j = 0; j < N; j++) H[j] = Gm[j];
  bool bad = false;
  double my_pivot = 1.0;
  static_for<N>([&](auto K) {
    constexpr int k = K;
    const double d = bc<k>(H[k]);
    bad = bad || !(d > 0.0);
    const double p = rcp_nr(d);
    const bool piv = lr == k;
    my_pivot = piv ? d : my_pivot;
    const double f = piv ? (1.0 - p) : H[k] * p;
    const double nf = -f;
    static_for<N>([&](auto J) {
      constexpr int j = J;
      if constexpr (j != k) fmac_bc<k, (j == (k == 0 ? 1 : 0))>(H[j], H[j], nf);
    });
    H[k] = piv ? p : nf;
  });
  const double rp = rsqrt_nr(my_pivot);
  const double c2 = row_sum(sel(var, rp, 0.0));
  double x = 0.0;
  {
    const double ng0 = -g0;
    double xa[3] = {0.0, 0.0, 0.0};
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(xa[j % 3], ng0, H[j]); });
    x = (xa[0] + xa[1]) + xa[2];
  }
  if (has_eq) {
    // all-zero column (every component exactly 0): the reference's dummy; otherwise a genuine equality
    const float nz = row_sum_f32(ne != 0.0 ? 1.0f : 0.0f);
    const bool dummy = !(nz > 0.0f);
    double za[3] = {0.0, 0.0, 0.0};
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(za[j % 3], ne, H[j]); });
    const double z = (za[0] + za[1]) + za[2];
    const double zn = row_sum(z * ne), zz = row_sum(z * z), nx = row_sum(ne * x);
    const double dinv = rcp_nr(zn);
    const double t2 = sel(fabs(zz) > eps, (-nx - ce0) * dinv, 0.0); // :176-179
    x += sel(dummy, 0.0, t2 * z);
    // H -= z z'/z'n (genuine), or only H[0][0] -= 1/G11 (dummy)
    const double vec = sel(dummy, 0.0, z * dinv), hc = sel(dummy, 0.0, -z);
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(H[j], vec, hc); });
    const double g11 = bc<0>(Gm[0]);
    H[0] = sel(dummy && lr == 0, H[0] - rcp_nr(g11), H[0]);
  }
  if (has_eq2) { // the second equality column: stepped onto and projected out of what the first has left
    double za[3] = {0.0, 0.0, 0.0};
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(za[j % 3], ne2, H[j]); });
    const double z = (za[0] + za[1]) + za[2];
    const double zn = row_sum(z * ne2), zz = row_sum(z * z), nx = row_sum(ne2 * x);
    const bool indep = fabs(zz) > eps && zn > 0.0;
    const double dinv = rcp_nr(indep ? zn : 1.0);
    x += sel(indep, (-nx - ce02) * dinv * z, 0.0);
    const double vec = sel(indep, z * dinv, 0.0), hc = sel(indep, -z, 0.0);
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(H[j], vec, hc); });
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);

  double Ns[N];
#pragma unroll
  for (int j = 0; j < N; j++) Ns[j] = 0.0;
  double u = 0.0;
  int idk = 0;
  unsigned used = 0;
  mask_t act_mask = 0, excl = 0;
  const mask_t one = 1;
  int q = 0, iters = 0, status = kStatusOk;
  const double psi_tol = (double)m * eps * c1 * c2 * 100.0;
  double rnorm2 = 1.0;
  bool done = skip, need_select = true, fresh = true;
  int ip = 0;
  double sp = 0.0, ucand = 0.0;
  if (bad && !skip) { status = kStatusNotPd; done = true; }

  const auto slacks = [&](double xx, double (&sl)[KC]) {
#pragma unroll
    for (int s = 0; s < KC; s++) sl[s] = b[s];
    static_for<N>([&](auto I) {
      constexpr int i = I;
      fmac_bc<i, i == 0>(sl[0], xx, a[0][i]);
#pragma unroll
      for (int s = 1; s < KC; s++) fmac_bc<i>(sl[s], xx, a[s][i]);
    });
  };

  for (int tick = 0; tick < 40 * kMaxOuter; tick++) {
    if (__all(done)) break;
    if (!done && need_select) {
      if (fresh) { iters++; excl = 0; }
      double sl[KC];
      slacks(x, sl);
      float viol = 0.0f;
#pragma unroll
      for (int s = 0; s < KC; s++) viol += v[s] ? (float)vmin(0.0, sl[s]) : 0.0f;
      const double psi = (double)row_sum_f32(viol);
      const mask_t blocked = act_mask | excl;
      double vv = sel(v[0] && !((blocked >> lr) & 1u) && sl[0] < 0.0, sl[0], inf);
      int which = 0;
#pragma unroll
      for (int s = 1; s < KC; s++) {
        const bool better = v[s] && !((blocked >> (lr + 16 * s)) & 1u) && sl[s] < 0.0 && sl[s] < vv;
        vv = sel(better, sl[s], vv);
        which = better ? s : which;
      }
      const double vbest = row_min(vv);
      const int wl = row_first(vv == vbest && vv < 0.0);
      const int sh = (threadIdx.x & 48) + (wl & 15);
      int wset = (int)((unsigned)(__ballot(which & 1) >> sh) & 1u);
      if constexpr (KC > 2) wset |= (int)((unsigned)(__ballot((which & 2) != 0) >> sh) & 1u) << 1;
      const bool feasible = fresh && (fabs(psi) <= psi_tol);     // QuadProg++.cc:246-250
      const bool stop = feasible || !(vbest < 0.0) || iters > kMaxOuter; // :271-274
      status = (stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      done = stop;
      ip = stop ? ip : (wl + 16 * wset);
      sp = sel(stop, sp, vbest);
      ucand = sel(stop, ucand, 0.0);
      need_select = stop;
    }
    if (!done) {
      const double npj = lr < N ? ct[N * ip + (lr < N ? lr : 0)] : 0.0;
      double za[3] = {0.0, 0.0, 0.0}, ra[3] = {0.0, 0.0, 0.0};
      static_for<N>([&](auto J) {
        constexpr int j = J;
        fmac_bc<j, j == 0>(za[j % 3], npj, H[j]);
        fmac_bc<j>(ra[j % 3], npj, Ns[j]);
      });
      const double z = (za[0] + za[1]) + za[2], r = (ra[0] + ra[1]) + ra[2];
      const bool slot = (used >> lr) & 1u;
      const double zn = row_sum(z * npj);
      const float zf = (float)z;
      const double zz = (double)row_sum_f32(zf * zf);
      const double ur = u * rcp_nr(r);
      const double ratio = sel(slot && r > 0.0, ur, inf);
      const double t1 = row_min(ratio);
      const int lpos = row_first(ratio == t1 && ratio < inf);
      const double t2v = -sp * rcp_nr(zn);
      // with n - (equality) constraints active the null space is empty and z is exactly 0 in the reference (J2 has
      // no columns); the explicit projector only leaves ~1e-7 of drift there, which must not pass for a direction
      const bool exhausted = q + (has_eq ? 1 : 0) + (has_eq2 ? 1 : 0) >= n_free;
      const double t2 = sel(!exhausted && fabs(zz) > eps && !(t2v < 0.0), t2v, inf);
      const double t = vmin(t1, t2);
      const bool infeasible = !(t < inf);                          // :339-344
      const bool dual_only = (t2 >= inf);
      const bool full = !infeasible && !dual_only && (t2 <= t1);   // :384
      const bool degenerate = full && !(zn > eps * eps * rnorm2);  // add_constraint failure (:392)
      const bool is_add = full && !degenerate;
      const bool is_drop = !infeasible && !full;
      if (infeasible) { status = kStatusInfeasible; done = true; }
      const double tp = (infeasible || dual_only || degenerate) ? 0.0 : t;
      const double td = (infeasible || degenerate) ? 0.0 : t;
      x += tp * z;
      u -= sel(slot, td * r, 0.0);
      ucand += td;
      sp += tp * zn;
      const int newlane = __ffs(~used & ((1u << N) - 1u)) - 1;
      const bool newslot = is_add && (lr == newlane);
      double vec = is_add ? z * rcp_nr(zn) : 0.0;
      double hc = is_add ? -z : 0.0;
      double nc = sel(newslot, 1.0, sel(is_add && slot, -r, 0.0));
      u = newslot ? ucand : u;
      idk = newslot ? ip : idk;
      used |= is_add ? (1u << newlane) : 0u;
      act_mask |= is_add ? (one << ip) : (mask_t)0;
      rnorm2 = is_add ? fmax(rnorm2, zn) : rnorm2;
      q += is_add ? 1 : 0;
      excl |= degenerate ? (one << ip) : (mask_t)0;
      need_select = need_select || full;
      fresh = is_add ? true : (degenerate ? false : fresh);
      if (is_drop) {
        if (lr == lpos) {
#pragma unroll
          for (int j = 0; j < N; j++) nrow[j] = Ns[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const double nt_me = lr < N ? nrow[lr < N ? lr : 0] : 0.0;
        double Gn = 0.0;
        static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(Gn, nt_me, Gm[j]); });
        const double einv = rcp_nr(row_sum(nt_me * Gn));
        double coef = 0.0;
        static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(coef, Gn, Ns[j]); });
        vec = nt_me;
        hc = nt_me * einv;
        nc = -coef * einv;
        const int drop_id = __shfl(idk, lpos, 16);
        act_mask &= ~(one << drop_id);
        used &= ~(1u << lpos);
        if (lr == lpos) u = 0.0;
        q--;
      }
      static_for<N>([&](auto J) {
        constexpr int j = J;
        fmac_bc<j, j == 0>(H[j], vec, hc);
        fmac_bc<j>(Ns[j], vec, nc);
      });
      if (is_drop && lr == lpos) {
#pragma unroll
        for (int j = 0; j < N; j++) Ns[j] = 0.0;
      }
    }
  }
  if (!done) status = kStatusMaxIter;
  // one refinement pass on the final working set (see balance_coop.hpp)
  if (status == kStatusOk && q > 0 && !skip) {
    if (lr < N) {
#pragma unroll
      for (int j = 0; j < N; j++) nst[N * lr + j] = Ns[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double NsT[N];
#pragma unroll
    for (int k = 0; k < N; k++) NsT[k] = lr < N ? nst[N * k + (lr < N ? lr : 0)] : 0.0;
    const bool myslot = (used >> lr) & 1u;
    double grad = g0;
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(grad, x, Gm[j]); });
    double corr = 0.0;
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(corr, grad, H[j]); });
    x -= corr;
    double sl[KC];
    slacks(x, sl);
    const int src = myslot ? (idk & 15) : 0;
    double sv = __shfl(sl[0], src, 16);
#pragma unroll
    for (int s = 1; s < KC; s++) {
      const double t = __shfl(sl[s], src, 16);
      sv = sel((idk >> 4) == s, t, sv);
    }
    const double rho = sel(myslot, -sv, 0.0);
    double dx = 0.0;
    static_for<N>([&](auto K) { constexpr int k = K; fmac_bc<k, k == 0>(dx, rho, NsT[k]); });
    x += dx;
  }
  // objective value at x: 1/2 x'Gx + g0'x  (the reference accumulates the same number step by step)
  {
    double gx = 0.0;
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(gx, x, Gm[j]); });
    const double fv = row_sum(sel(var, x * (0.5 * gx + g0), 0.0));
    f_out = status == kStatusInfeasible ? inf : fv;
  }
  x_out = x;
  return status;
}

// two inequalities per lane (m <= 24): the shape qlamd_qp_solve_batch uses
template <int N>
__device__ __forceinline__ int qp_coop(const double (&Gm)[N], double g0, int n, int m, bool has_eq, double ne, double ce0,
                                       const double (&a0)[N], double b0, bool v0, const double (&a1)[N], double b1, bool v1,
                                       bool skip, double *lds_row, double &x_out, double &f_out) {
  double a[2][N];
#pragma unroll
  for (int i = 0; i < N; i++) { a[0][i] = a0[i]; a[1][i] = a1[i]; }
  const double b[2] = {b0, b1};
  const bool v[2] = {v0, v1};
  return qp_coop_impl<N, 2>(Gm, g0, n, n, m, has_eq, ne, ce0, a, b, v, skip, lds_row, x_out, f_out);
}

} // namespace coop
} // namespace qlamd
