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
  enum { kRows = KC == 2 ? 24 : 16 * KC, kCt = 0, kNrow = kRows * N, kNst = kRows * N + N, kLatch = kRows * N + N + N * N,
         kZero = kLatch + 48, kTotal = kLatch + 50 }; // kLatch: what a finished row keeps for the epilogue (ghost rows, force_qp_coop.hpp); kZero: a zero
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
                                            double ne2 = 0.0, double ce02 = 0.0, const double *eq_rows = nullptr,
                                            const double *eq_rhs = nullptr, int p_rows = 0, int m_tol = -1,
                                            double *eq_store = nullptr, int *iters_out = nullptr) {
  typedef QpCoopLds<N, KC> L;
  typedef typename std::conditional<(KC > 2), unsigned long long, unsigned>::type mask_t;
  const int lr = threadIdx.x & 15;
  const bool var = lr < n;
  const double eps = 2.220446049250313e-16;
  const double inf = INFINITY;
  double *ct = lds_row + L::kCt, *nrow = lds_row + L::kNrow, *nst = lds_row + L::kNst, *latched = lds_row + L::kLatch;

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
  bool eq2_dependent = false;
  if (has_eq2) { // the second equality column: stepped onto and projected out of what the first has left
    double za[3] = {0.0, 0.0, 0.0};
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(za[j % 3], ne2, H[j]); });
    const double z = (za[0] + za[1]) + za[2];
    const double zn = row_sum(z * ne2), zz = row_sum(z * z), nx = row_sum(ne2 * x);
    const bool indep = fabs(zz) > eps && zn > 0.0;
    eq2_dependent = !indep; // all-zero or in the span of the first: left out, reported through the status
    const double dinv = rcp_nr(indep ? zn : 1.0);
    x += sel(indep, (-nx - ce02) * dinv * z, 0.0);
    const double vec = sel(indep, z * dinv, 0.0), hc = sel(indep, -z, 0.0);
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(H[j], vec, hc); });
  }
  // Any number of equality ROWS  c_r'x = rhs_r  (eq_rows: this problem's [p_rows][n], row-major; the form
  // ooqpei::QuadraticProblemFormulation::solve takes its C, c in): each is stepped onto and projected out of what the
  // rows before it have left, exactly as the columns above.  A row that leaves nothing -- all-zero, or in the span of
  // the earlier ones -- is skipped when the point already satisfies it (the reference's first pass hands over 3 nS
  // zero rows with zero right-hand sides, ContactForceDistribution.cpp:364-366) and makes the problem infeasible otherwise.
  int neq_rows = 0;
  bool eq_inconsistent = false;
  for (int r = 0; r < p_rows; r++) {
    const double ner = var ? eq_rows[(size_t)r * n + lr] : 0.0;
    const double rhs = eq_rhs[r];
    double za[3] = {0.0, 0.0, 0.0};
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(za[j % 3], ner, H[j]); });
    const double z = (za[0] + za[1]) + za[2];
    const double zn = row_sum(z * ner), zz = row_sum(z * z), nx = row_sum(ner * x), ax = row_sum(fabs(ner * x));
    const bool indep = fabs(zz) > eps && zn > 0.0 && neq_rows + (has_eq ? 1 : 0) + ((has_eq2 && !eq2_dependent) ? 1 : 0) < n_free;
    const double resid = rhs - nx;
    eq_inconsistent = eq_inconsistent || (!indep && fabs(resid) > 1e-9 * (1.0 + fabs(rhs) + ax));
    neq_rows += indep ? 1 : 0;
    const double dinv = rcp_nr(indep ? zn : 1.0);
    x += sel(indep, resid * dinv * z, 0.0);
    const double vec = sel(indep, z * dinv, 0.0), hc = sel(indep, -z, 0.0);
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(H[j], vec, hc); });
    if (eq_store && lr < N) eq_store[r * N + lr] = vec; // the step direction of row r per unit of its residual
  }
  if (eq_store && p_rows > 0) {
    // One sweep of refinement over the rows just projected out.  Direction r lies in the null space of rows 0..r-1 only up
    // to the rounding of an explicit projector (times the conditioning of G: 3e5 for the force problem), so the rows
    // taken first have drifted by the time the last one is in; stepping again along the stored directions, in the same
    // order, puts every row back on its right-hand side (row r's step leaves rows 0..r-1 where the sweep has put them).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int r = 0; r < p_rows; r++) {
      const double ner = var ? eq_rows[(size_t)r * n + lr] : 0.0;
      const double resid = eq_rhs[r] - row_sum(ner * x);
      x += lr < N ? resid * eq_store[r * N + lr] : 0.0;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);

  // Active-set loop in the form of balance_coop.hpp (see the notes there): selection of the next violated constraint
  // by 32-bit keys (slack in single precision, low six bits = lane and which of my KC rows) with one DPP max per
  // level, its slack fetched from its lane, the feasibility sum only when the worst slack is inside the tolerance;
  // the selection sits in the shadow of the rank-one update; passes in which every live row adds run in an inner
  // loop without predication, everything else goes through the general tail.
  double Ns[N];
#pragma unroll
  for (int j = 0; j < N; j++) Ns[j] = 0.0;
  double u = 0.0;
  int idk = 0;
  unsigned used = 0;
  mask_t act_mask = 0, excl = 0;
  const mask_t one = 1;
  int q = 0, iters = 0, status = kStatusOk;
  const double psi_tol = (double)(m_tol >= 0 ? m_tol : m) * eps * c1 * c2 * 100.0; // m_tol: rows that exist, when m counts slots
  double rnorm2 = 1.0;
  // Finished rows ride along as ghost rows (force_qp_coop.hpp: full EXEC, scalar-mask loop control): their candidate
  // normal is 0 (nlim: the lanes that take a component of the chosen normal, none on a ghost row) and their slot of the drop
  // export reads a zero, so z = r = n~ = 0 and the rank-one updates add exact zeros; z'n_p and n~'G n~ are biased to 1 (zb);
  // their results wait in LDS (latched).
  unsigned long long done_m = 0ull;
  int nlim = N;
  int nt_slot = L::kNrow + (lr < N ? lr : 0);
  double zb = 0.0;
  int ip = 0;
  double sp = 0.0, ucand = 0.0, npj = 0.0;
  bool fin_before = skip; // rows that never enter the loop
  if (bad && !skip) { status = kStatusNotPd; fin_before = true; }
  if (eq_inconsistent && !bad && !skip) { status = kStatusInfeasible; fin_before = true; }
  lds_row[L::kZero] = 0.0;
  const int neq = (has_eq ? 1 : 0) + ((has_eq2 && !eq2_dependent) ? 1 : 0) + neq_rows; // equalities that took a dimension
  const unsigned lanebit = 1u << lr;
  const int row_addr = ((int)threadIdx.x & 48) << 2;
  const int vlane = lr < N ? lr : 0;

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
  const auto umax_dpp = [](unsigned k, auto Ctrl) -> unsigned {
    constexpr int ctrl = decltype(Ctrl)::value;
    const unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, ctrl, 0xF, 0xF, true);
    return k > o ? k : o;
  };
  double vec = 0.0, hc = 0.0, nc = 0.0;
  // kMode 0: before the first step (no update, every live row selects); 1: general (rows in `resel` select, `fresh`
  // after an add); 2: every live row has just added a constraint
  const auto update_and_select = [&](auto Mode, bool resel, bool fresh, bool finished = false) -> bool {
    constexpr int kMode = decltype(Mode)::value;
    if constexpr (kMode == 1) {
      iters += (resel && fresh) ? 1 : 0;
      excl = (resel && fresh) ? (mask_t)0 : excl;
    } else {
      iters += 1;
      excl = 0;
    }
    const mask_t avail = ~(act_mask | excl);
    double sl[KC];
    slacks(x, sl);
    unsigned key = 0u;
    double myv = sl[0];
#pragma unroll
    for (int s = 0; s < KC; s++) {
      unsigned k = __float_as_uint((float)sl[s]);
      k = (v[s] && ((avail >> (lr + 16 * s)) & 1u) && sl[s] < 0.0) ? ((k & ~63u) | (unsigned)(lr | (s << 4))) : 0u;
      myv = k > key ? sl[s] : myv;
      key = k > key ? k : key;
    }
    if constexpr (kMode != 0) {
      static_for<N>([&](auto J) {
        constexpr int j = J;
        fmac_bc<j, j == 0>(H[j], vec, hc);
        fmac_bc<j>(Ns[j], vec, nc);
      });
    }
    key = umax_dpp(key, std::integral_constant<int, 0x128>{});
    key = umax_dpp(key, std::integral_constant<int, 0x124>{});
    key = umax_dpp(key, std::integral_constant<int, 0x122>{});
    key = umax_dpp(key, std::integral_constant<int, 0x121>{});
    const int wl = (int)key & 15;
    const int addr = row_addr + (wl << 2);
    const int vlo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(myv));
    const int vhi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(myv));
    const int key_ip = wl + 16 * (((int)key >> 4) & 3);
    const double np_tab = ct[N * key_ip + vlane];
    const double np_new = lr < nlim ? np_tab : 0.0;
    const bool any = (int)key < 0;
    const double vsel = __hiloint2double(vhi, vlo);
    bool feasible = false;
    const bool close = (kMode != 1 || (resel && fresh)) && any && !(vsel < -psi_tol);
    if (__builtin_amdgcn_ballot_w64(close) != 0ull) { // QuadProg++.cc:246-250
      float viol = 0.0f;
#pragma unroll
      for (int s = 0; s < KC; s++) viol += v[s] ? (float)vmin(0.0, sl[s]) : 0.0f;
      const double psi = (double)row_sum_f32(viol);
      feasible = close && (fabs(psi) <= psi_tol);
    }
    const bool stop = !any || feasible || iters > kMaxOuter; // :271-274
    bool fin = stop;
    if constexpr (kMode == 1) {
      status = (resel && stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      fin = (resel && stop) || finished;
      const bool take = resel && !stop;
      ip = take ? key_ip : ip;
      sp = sel(take, vsel, sp);
      ucand = sel(take, 0.0, ucand);
      npj = sel(take, np_new, npj);
    } else { // (a ghost row's status is latched: what this writes to it is never read)
      status = (stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      ip = key_ip; sp = vsel; ucand = 0.0; npj = np_new;
    }
    return fin;
  };
  const auto update_only = [&]() {
    static_for<N>([&](auto J) {
      constexpr int j = J;
      fmac_bc<j, j == 0>(H[j], vec, hc);
      fmac_bc<j>(Ns[j], vec, nc);
    });
  };
  // called where control is uniform, with the rows that have just finished: they latch their results and turn into ghosts
  const auto note_finished = [&](bool fin) {
    const unsigned long long fin_m = __builtin_amdgcn_ballot_w64(fin);
    const unsigned long long newly_m = fin_m & ~done_m;
    done_m |= fin_m;
    if (newly_m != 0ull) {
      if (__builtin_amdgcn_inverse_ballot_w64(newly_m)) {
        latched[lr] = x;
        reinterpret_cast<int2 *>(latched + 16)[lr] = make_int2((int)used, idk);
        reinterpret_cast<int2 *>(latched + 32)[lr] = make_int2(q | (iters << 8), status);
        nlim = 0; nt_slot = L::kZero; zb = 0.0625; npj = 0.0;
      }
    }
  };
  double drop_einv = 0.0;
  const auto drop_vectors = [&](int lpos) {
    if (lr == lpos) {
#pragma unroll
      for (int j = 0; j < N; j++) nrow[j] = Ns[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const double nt_me = lr < N ? lds_row[nt_slot] : 0.0;
    const int drop_id = __shfl(idk, lpos, 16);
    double Gn = 0.0;
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(Gn, nt_me, Gm[j]); });
    const double einv = rcp_nr1(row_sum(fma(nt_me, Gn, zb)));
    drop_einv = einv;
    double coef = 0.0;
    static_for<N>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(coef, Gn, Ns[j]); });
    vec = nt_me;
    hc = nt_me * einv;
    nc = sel(lr == lpos, -1.0, -coef * einv); // exactly -1 on the dropped slot: its row becomes exactly 0 (force_qp_coop.hpp)
    return drop_id;
  };

  {
    // lanes that are not here count as finished (ballots never see them); rows that failed before the loop finish now
    done_m = ~__builtin_amdgcn_ballot_w64(true);
    const bool fin0 = update_and_select(std::integral_constant<int, 0>{}, true, true);
    note_finished(fin0 || fin_before);
  }

  // Loop as in force_qp_coop.hpp: scalar masks of the rows that have finished / add / drop in the pass at hand; an inner
  // loop of the passes in which every live row adds; otherwise the blocked rows drop inside the pass (straight drop path,
  // step lengths again from the continued directions) until every live row can add, and then all add together; the
  // general predicated form for a row that does neither.
  double z = 0.0, r = 0.0, zn = 0.0, zinv = 0.0, t = 0.0, tl1 = 0.0, tl2 = 0.0, ratio = 0.0;
  const auto in = [](unsigned long long mm) -> bool { return __builtin_amdgcn_inverse_ballot_w64(mm); };
  const auto directions = [&]() {
    double za[3] = {0.0, 0.0, 0.0}, ra[3] = {0.0, 0.0, 0.0};
    static_for<N>([&](auto J) {
      constexpr int j = J;
      fmac_bc<j, j == 0>(za[j % 3], npj, H[j]);
      fmac_bc<j>(ra[j % 3], npj, Ns[j]);
    });
    z = (za[0] + za[1]) + za[2];
    r = (ra[0] + ra[1]) + ra[2];
    zn = row_sum(fma(z, npj, zb));
  };
  // step lengths (QuadProg++.cc:304-331); returns whether the pass is a full step that adds the candidate
  const auto step_lengths = [&]() -> bool {
    const bool slot = (used & lanebit) != 0u;
    const float zf = (float)z;
    const double zz = (double)row_sum_f32(zf * zf);
    const double ur = u * rcp_nr1(r);
    ratio = sel(slot && r > 0.0, ur, inf);
    tl1 = row_min(ratio);
    zinv = rcp_nr(zn);
    const double t2v = -sp * zinv;
    // with n - (equality) constraints active the null space is empty and z is exactly 0 in the reference (J2 has
    // no columns); the explicit projector only leaves ~1e-7 of drift there, which must not pass for a direction
    const bool exhausted = q + neq >= n_free;
    tl2 = sel((int)(!exhausted) & (int)(fabs(zz) > eps) & (int)(!(t2v < 0.0)), t2v, inf);
    t = vmin(tl1, tl2);
    return (tl2 < inf) && (tl2 <= tl1) && (zn > eps * eps * rnorm2); // full step (:384), add_constraint succeeds (:392)
  };
  const auto add_step = [&]() -> bool {
    x += t * z;
    u = fma(-t, r, u);
    const int newlane = __ffs(~used & ((1u << N) - 1u)) - 1;
    const bool newslot = lr == newlane;
    vec = z * zinv;
    hc = -z;
    nc = sel(newslot, 1.0, -r);
    u = sel(newslot, ucand + t, u);
    idk = newslot ? ip : idk;
    used |= 1u << newlane;
    act_mask |= one << ip;
    rnorm2 = vmax(rnorm2, zn);
    q += 1;
    return update_and_select(std::integral_constant<int, 2>{}, true, true);
  };
  // a partial step (t1 < t2), or a dual step only when t2 is infinite: the blocking constraint leaves the working set and the
  // same candidate continues; its directions follow from the rank-one update just made (force_qp_coop.hpp)
  const auto drop_step = [&]() {
    const double tp = (tl2 >= inf) ? 0.0 : t;
    x += tp * z;
    u = fma(-t, r, u);
    ucand += t;
    sp += tp * zn;
    const int lpos = row_first(ratio == tl1 && ratio < inf);
    const double r_lpos = __shfl(r, lpos, 16);
    const int drop_id = drop_vectors(lpos);
    act_mask &= ~(one << drop_id);
    used &= ~(1u << lpos);
    q--;
    update_only();
    z = fma(hc, r_lpos, z);
    r = fma(nc, r_lpos, r);
    zn = fma(r_lpos * r_lpos, drop_einv, zn);
  };
  const auto general_pass = [&](bool is_add) -> bool {
    const bool infeasible = !(t < inf);                            // :339-344
    const bool dual_only = (tl2 >= inf);
    const bool full = !infeasible && !dual_only && (tl2 <= tl1);   // :384
    const bool degenerate = full && !is_add;                       // add_constraint failure (:392)
    const bool is_drop = !infeasible && !full;
    if (infeasible) status = kStatusInfeasible;
    const double tp = (infeasible || dual_only || degenerate) ? 0.0 : t;
    const double td = (infeasible || degenerate) ? 0.0 : t;
    x += tp * z;
    u = fma(-td, r, u);
    ucand += td;
    sp += tp * zn;
    const int newlane = __ffs(~used & ((1u << N) - 1u)) - 1;
    const bool newslot = is_add && (lr == newlane);
    vec = is_add ? z * zinv : 0.0;
    hc = is_add ? -z : 0.0;
    nc = sel(newslot, 1.0, sel(is_add, -r, 0.0));
    u = newslot ? ucand : u;
    idk = newslot ? ip : idk;
    used |= is_add ? (1u << newlane) : 0u;
    act_mask |= is_add ? (one << ip) : (mask_t)0;
    rnorm2 = is_add ? vmax(rnorm2, zn) : rnorm2;
    q += is_add ? 1 : 0;
    excl |= degenerate ? (one << ip) : (mask_t)0;
    int lpos = 16;
    if (is_drop) {
      lpos = row_first(ratio == tl1 && ratio < inf);
      const int drop_id = drop_vectors(lpos);
      act_mask &= ~(one << drop_id);
      used &= ~(1u << lpos);
      q--;
    }
    return update_and_select(std::integral_constant<int, 1>{}, full, is_add, infeasible);
  };
  if (~done_m != 0ull) {
    for (;;) {
      unsigned long long add_m = 0ull;
      for (;;) { // passes in which every live row adds
        directions();
        add_m = __builtin_amdgcn_ballot_w64(step_lengths());
        if (~(add_m | done_m) != 0ull) break;
        note_finished(add_step());
        if (~done_m == 0ull) break;
      }
      if (~done_m == 0ull) break;
      unsigned long long gen_m = 0ull; // rows that took the general form in this pass
      bool fin = false;
      for (;;) {
        const unsigned long long live_m = ~done_m & ~gen_m;
        // a dual step only (t2 infinite) with a blocking constraint is a drop as well; without one the problem is infeasible
        const unsigned long long drop_m = __builtin_amdgcn_ballot_w64(tl1 < tl2) & live_m & ~add_m;
        const unsigned long long other_m = live_m & ~add_m & ~drop_m;
        if (other_m != 0ull) {
          if (in(other_m)) fin = general_pass(false);
          gen_m |= other_m;
        }
        if (drop_m == 0ull) break;
        if (in(drop_m | done_m)) drop_step();
        add_m = __builtin_amdgcn_ballot_w64(step_lengths());
      }
      const unsigned long long adders_m = add_m & ~done_m & ~gen_m;
      if (adders_m != 0ull) {
        bool fin_add = false;
        if (in(adders_m | done_m)) fin_add = add_step();
        fin = fin || fin_add;
      }
      note_finished(fin);
      if (~done_m == 0ull) break;
    }
  }
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    x = latched[lr];
    const int2 la = reinterpret_cast<const int2 *>(latched + 16)[lr], lb = reinterpret_cast<const int2 *>(latched + 32)[lr];
    used = (unsigned)la.x; idk = la.y; q = lb.x & 255; status = lb.y;
    if (iters_out) *iters_out = lb.x >> 8;
  }
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
  return (status == kStatusOk && eq2_dependent) ? kStatusDependentEquality : status;
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
