// Lane-cooperative pose optimisation (BASELINE config 5): 16 lanes per problem, 4 problems per wavefront.
// Device-only; included by balance_kernel.hip after balance_coop.hpp (whose DPP helpers it uses).
//
// Same algorithm as pose_sqp / pose_sqp6 in pose_core.hpp (PoseOptimizationSQP.cpp:58-111,
// sequencequadraticproblemsolver.cpp:18-102, the objective / constraint classes cited there), restated so that one
// problem's work is spread over a DPP row instead of running as one lane's serial instruction stream:
//   lanes 0..5   variable lanes: component i of x, g0, z; row i of the 6x6 Hessian G and of the operator H;
//                also slot lanes: row k of N*, multiplier u_k, constraint id of active-set slot k
//   lanes 8..15  constraint lanes: constraint j = lane - 8 (0..3 support-region half-spaces, 4..7 limb-length
//                bounds of the stance legs in iteration order): its normal a_j[0..5], its bound, its slack
//   everything that is per problem (pose, rotation, sums over legs) is replicated on the 16 lanes.
//
// Linearisation.  The reference builds the rotational Hessian block from four 3x3 products of skew matrices per
// leg; with skew(a) skew(b) = b a' - (a.b) I they collapse to
//   1/2 (T1 + T2 - T3 - T4) = 1/2 (Pd e' + e Pd') - (e.Pd) I,   e = p - f_k,  Pd = R d_k,
// (and the same with Pr, p - c for the centre-of-mass term), which is what is evaluated here.
//
// Inner QP.  Goldfarb-Idnani with explicit operators as in balance_coop.hpp.  The reference always passes one
// all-zero equality column.  In QuadProg++ that column consumes the first column of J = L^-T without
// moving x, which is exactly an equality constraint with normal G e_1 (the first column of G) that holds at the
// unconstrained minimiser x0: the projector starts as H0 = G^-1 - e_1 e_1' / G_11 instead of G^-1 (checked against
// the reference solver on random QPs to 5e-13, tests/test_oracle_quadprog.py).  Its multiplier is never needed,
// so no slot is spent on it.
#pragma once

#include "balance_coop.hpp"
#include "pose_core.hpp"

namespace qlamd {
namespace coop {

constexpr int kPoseCoopRows = 4;                 // problems per wavefront
constexpr int kPoseCoopLdsDoubles = 8 * 6 + 6 + 6 * 6; // per problem: normals by constraint, one N* row, N* (refinement)

// ---- the 6-variable QP ------------------------------------------------------------------------------------
// Gm: row lr of G on variable lanes (0 elsewhere); g0: component lr; a, bci0, cvalid: this constraint lane's
// normal (CI column), ci0 and presence; m: number of inequality constraints present.  Returns the status;
// x_out = component lr of the minimiser on variable lanes.
// n: number of real variables (6 for the SQP step; 3 for the position-only QP, whose rows 3..5 of G are identity
// padding -- those variables stay 0 and do not enter the traces behind the termination tolerance).
// kDiagTop: the leading 3x3 block of G is diagonal (the position block of the pose Hessian, 2 (nl I + w diag(1, 1, 0)), and
// the position-only QP, whose G is diagonal altogether): its three pivots are known without elimination.
template <bool kDiagTop = false>
__device__ __forceinline__ int qp6_coop(const double (&Gm)[6], double g0, const double (&a)[6], double bci0, bool cvalid,
                                        int m, bool dummy_eq, bool skip, double *lds_row, double &x_out, int n = 6) {
  const int lr = threadIdx.x & 15;
  const bool var = lr < 6;
  const bool real = lr < n;
  const int cj = lr - 8; // constraint id on constraint lanes
  const double eps = 2.220446049250313e-16;
  const double inf = INFINITY;
  double *ct = lds_row, *nrow = lds_row + 48, *nst = lds_row + 54;

  // normals by constraint into LDS: variable lane i later reads a_p[i] = ct[6 p + i]
  if (lr >= 8) {
#pragma unroll
    for (int i = 0; i < 6; i++) ct[6 * cj + i] = a[i];
  }
  // c1 = trace(G)
  double diag = 0.0;
#pragma unroll
  for (int j = 0; j < 6; j++) diag = sel(lr == j, Gm[j], diag);
  const double c1 = row_sum(sel(real, diag, 0.0));
  // H = G^-1 by Gauss-Jordan, row per lane (non-variable lanes carry zero rows)
  double H[6];
#pragma unroll
  for (int j = 0; j < 6; j++) H[j] = Gm[j];
  bool bad = false;
  double my_pivot = 1.0;
  // as in force_qp_coop.hpp: the column of the NEXT pivot is updated first, so that its reciprocal (seed + one Newton
  // step: the refinement and the next SQP iteration work on G itself) is under way while the other columns are updated
  constexpr int kFirst = kDiagTop ? 3 : 0;
  if constexpr (kDiagTop) {
    // rows 0..2 do not meet in the first three columns, so their pivots are the diagonal entries as they stand, the three
    // reciprocals run side by side, and the three eliminations touch columns 3..5 (and their own column) only: nine
    // broadcast-FMAs in any order instead of three rounds of the serial chain pivot -> reciprocal -> factor -> update
    double nfk[3], pk[3];
    static_for<3>([&](auto K) {
      constexpr int k = K;
      const double dk = bc<k>(H[k]);
      bad = bad || !(dk > 0.0);
      pk[k] = rcp_nr1(dk);
      const bool piv = lr == k;
      my_pivot = piv ? dk : my_pivot;
      nfk[k] = -(piv ? (1.0 - pk[k]) : H[k] * pk[k]);
    });
    static_for<3>([&](auto K) {
      constexpr int k = K;
      fmac_bc<k, k == 0>(H[3], H[3], nfk[k]);
      fmac_bc<k>(H[4], H[4], nfk[k]);
      fmac_bc<k>(H[5], H[5], nfk[k]);
    });
    static_for<3>([&](auto K) { constexpr int k = K; H[k] = lr == k ? pk[k] : nfk[k]; });
  }
  double d = bc<kFirst>(H[kFirst]);
  static_for<6 - kFirst>([&](auto K) {
    constexpr int k = kFirst + decltype(K)::value;
    bad = bad || !(d > 0.0);
    const double p = rcp_nr1(d);
    const bool piv = lr == k;
    my_pivot = piv ? d : my_pivot;
    const double f = piv ? (1.0 - p) : H[k] * p;
    const double nf = -f;
    if constexpr (k < 5) {
      fmac_bc<k, true>(H[k + 1], H[k + 1], nf);
      d = bc<k + 1>(H[k + 1]);
    }
    static_for<6>([&](auto J) {
      constexpr int j = J;
      if constexpr (j != k && j != k + 1) fmac_bc<k, (k == 5 && j == 0)>(H[j], H[j], nf);
    });
    H[k] = piv ? p : nf;
  });
  QL_STAMP(23);
  const double rp = rsqrt_nr(my_pivot);
  const double c2 = row_sum(sel(real, rp, 0.0));
  // x0 = -G^-1 g0 (the equality below holds there by construction)
  double x = 0.0;
  {
    const double ng0 = -g0;
    double xa[2] = {0.0, 0.0};
    static_for<6>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(xa[j & 1], ng0, H[j]); });
    x = xa[0] + xa[1];
  }
  if (dummy_eq) { // H0 = G^-1 - e1 e1'/G11: only element (0,0) changes
    const double g11 = bc<0>(Gm[0]);
    H[0] = sel(lr == 0, H[0] - rcp_nr(g11), H[0]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F); // the normals are in LDS

  // Active-set loop in the form of balance_coop.hpp (see the notes there): key selection on the constraint lanes,
  // the chosen slack fetched from its lane, lazy feasibility sum, selection in the shadow of the rank-one update,
  // unpredicated inner loop for the passes in which every live row adds.
  double Ns[6];
#pragma unroll
  for (int j = 0; j < 6; j++) Ns[j] = 0.0;
  double u = 0.0;
  int idk = 0;
  unsigned used = 0, act_mask = 0, excl = 0;
  int q = 0, iters = 0, status = kStatusOk;
  const double psi_tol = (double)m * eps * c1 * c2 * 100.0;
  double rnorm2 = 1.0;
  bool done = skip;
  int ip = 0;
  double sp = 0.0, ucand = 0.0, npj = 0.0;
  if (bad && !skip) { status = kStatusNotPd; done = true; }
  const unsigned lanebit = 1u << lr;
  const int row_addr = ((int)threadIdx.x & 48) << 2;
  const int vlane = var ? lr : 0;
  const bool mine = cvalid && lr >= 8;
  const unsigned cbit = mine ? (1u << (cj & 7)) : 0u;

  const auto umax_dpp = [](unsigned k, auto Ctrl) -> unsigned {
    constexpr int ctrl = decltype(Ctrl)::value;
    const unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, ctrl, 0xF, 0xF, true);
    return k > o ? k : o;
  };
  double vec = 0.0, hc = 0.0, nc = 0.0;
  const auto update_and_select = [&](auto Mode, bool resel, bool fresh) {
    constexpr int kMode = decltype(Mode)::value;
    if constexpr (kMode == 1) {
      iters += (resel && fresh) ? 1 : 0;
      excl = (resel && fresh) ? 0u : excl;
    } else {
      iters += 1;
      excl = 0u;
    }
    const unsigned avail = ~(act_mask | excl);
    // slack of my constraint: s_j = a_j'x + ci0_j
    double s = bci0, s1 = 0.0;
    static_for<3>([&](auto I) { constexpr int i = I; fmac_bc<2 * i, i == 0>(s, x, a[2 * i]); fmac_bc<2 * i + 1>(s1, x, a[2 * i + 1]); });
    s += s1;
    unsigned key = __float_as_uint((float)s);
    key = ((avail & cbit) != 0u && s < 0.0) ? ((key & ~15u) | (unsigned)lr) : 0u;
    if constexpr (kMode != 0) {
      static_for<6>([&](auto J) {
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
    const int vlo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(s));
    const int vhi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(s));
    const int key_ip = (wl - 8) & 7;
    const double np_tab = ct[6 * key_ip + vlane];
    const double np_new = var ? np_tab : 0.0;
    const bool any = (int)key < 0;
    const double vsel = __hiloint2double(vhi, vlo);
    bool feasible = false;
    const bool close = (kMode != 1 || (resel && fresh)) && any && !(vsel < -psi_tol);
    if (__builtin_amdgcn_ballot_w64(close) != 0ull) { // QuadProg++.cc:246-250
      const float viol = mine ? (float)vmin(0.0, s) : 0.0f;
      const double psi = (double)row_sum_f32(viol);
      feasible = close && (fabs(psi) <= psi_tol);
    }
    const bool stop = !any || feasible || iters > kMaxOuter; // :271-274
    if constexpr (kMode == 1) {
      status = (resel && stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      done = done || (resel && stop);
      const bool take = resel && !stop;
      ip = take ? key_ip : ip;
      sp = sel(take, vsel, sp);
      ucand = sel(take, 0.0, ucand);
      npj = sel(take, np_new, npj);
    } else {
      status = (!done && stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      done = done || stop;
      ip = key_ip; sp = vsel; ucand = 0.0; npj = np_new;
    }
  };
  const auto drop_vectors = [&](int lpos) {
    if (lr == lpos) {
#pragma unroll
      for (int j = 0; j < 6; j++) nrow[j] = Ns[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const double nt_me = var ? nrow[vlane] : 0.0;
    const int drop_id = __shfl(idk, lpos, 16);
    double Gn = 0.0;
    static_for<6>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(Gn, nt_me, Gm[j]); });
    const double einv = rcp_nr1(row_sum(nt_me * Gn));
    double coef = 0.0;
    static_for<6>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(coef, Gn, Ns[j]); });
    vec = nt_me;
    hc = nt_me * einv;
    nc = sel(lr == lpos, -1.0, -coef * einv); // exactly -1 on the dropped slot: its row becomes exactly 0 (force_qp_coop.hpp)
    return drop_id;
  };

  update_and_select(std::integral_constant<int, 0>{}, true, true);
  QL_STAMP(24);

  for (;;) {
    double z = 0.0, r = 0.0, zn = 0.0, zinv = 0.0, t = 0.0, tl1 = 0.0, tl2 = 0.0, ratio = 0.0;
    bool is_add = false;
    while (!done) {
      double za[2] = {0.0, 0.0}, ra[2] = {0.0, 0.0};
      static_for<6>([&](auto J) {
        constexpr int j = J;
        fmac_bc<j, j == 0>(za[j & 1], npj, H[j]);
        fmac_bc<j>(ra[j & 1], npj, Ns[j]);
      });
      z = za[0] + za[1];
      r = ra[0] + ra[1];
      const bool slot = (used & lanebit) != 0u;
      zn = row_sum(z * npj);
      const float zf = (float)z;
      const double zz = (double)row_sum_f32(zf * zf);
      // step lengths, QuadProg++.cc:304-331
      const double ur = u * rcp_nr1(r);
      ratio = sel(slot && r > 0.0, ur, inf);
      tl1 = row_min(ratio);
      zinv = rcp_nr(zn);
      const double t2v = -sp * zinv;
      const bool exhausted = q + (dummy_eq ? 1 : 0) >= n; // empty null space: z is exactly 0 in the reference
      tl2 = sel((int)(!exhausted) & (int)(fabs(zz) > eps) & (int)(!(t2v < 0.0)), t2v, inf);
      t = vmin(tl1, tl2);
      is_add = (tl2 < inf) && (tl2 <= tl1) && (zn > eps * eps * rnorm2); // full step (:384), add_constraint succeeds (:392)
      if (__builtin_amdgcn_ballot_w64(!is_add) != 0ull) break;
      x += t * z;
      u = fma(-t, r, u);
      const int newlane = __ffs(~used & 0x3Fu) - 1;
      const bool newslot = lr == newlane;
      vec = z * zinv;
      hc = -z;
      nc = sel(newslot, 1.0, -r);
      u = sel(newslot, ucand + t, u);
      idk = newslot ? ip : idk;
      used |= 1u << newlane;
      act_mask |= 1u << ip;
      rnorm2 = vmax(rnorm2, zn);
      q += 1;
      update_and_select(std::integral_constant<int, 2>{}, true, true);
    }
    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
    if (!done) {
      const bool infeasible = !(t < inf);                            // :339-344
      const bool dual_only = (tl2 >= inf);
      const bool full = !infeasible && !dual_only && (tl2 <= tl1);   // :384
      const bool degenerate = full && !is_add;                       // add_constraint failure (:392)
      const bool is_drop = !infeasible && !full;
      if (infeasible) { status = kStatusInfeasible; done = true; }
      const double tp = (infeasible || dual_only || degenerate) ? 0.0 : t;
      const double td = (infeasible || degenerate) ? 0.0 : t;
      x += tp * z;
      u = fma(-td, r, u);
      ucand += td;
      sp += tp * zn;
      const int newlane = __ffs(~used & 0x3Fu) - 1;
      const bool newslot = is_add && (lr == newlane);
      vec = is_add ? z * zinv : 0.0;
      hc = is_add ? -z : 0.0;
      nc = sel(newslot, 1.0, sel(is_add, -r, 0.0));
      u = newslot ? ucand : u;
      idk = newslot ? ip : idk;
      used |= is_add ? (1u << newlane) : 0u;
      act_mask |= is_add ? (1u << ip) : 0u;
      rnorm2 = is_add ? vmax(rnorm2, zn) : rnorm2;
      q += is_add ? 1 : 0;
      excl |= degenerate ? (1u << ip) : 0u;
      int lpos = 16;
      if (is_drop) {
        lpos = row_first(ratio == tl1 && ratio < inf);
        const int drop_id = drop_vectors(lpos);
        act_mask &= ~(1u << drop_id);
        used &= ~(1u << lpos);
        q--;
      }
      update_and_select(std::integral_constant<int, 1>{}, full, is_add);
    }
  }
  QL_STAMP(25);
  // one refinement pass on the final working set (see balance_coop.hpp)
  if (status == kStatusOk && q > 0 && !skip) {
    if (lr < 6) {
#pragma unroll
      for (int j = 0; j < 6; j++) nst[6 * lr + j] = Ns[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double NsT[6];
#pragma unroll
    for (int k = 0; k < 6; k++) NsT[k] = var ? nst[6 * k + lr] : 0.0;
    const bool myslot = (used >> lr) & 1u;
    double grad = g0;
    static_for<6>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(grad, x, Gm[j]); });
    double corr = 0.0;
    static_for<6>([&](auto J) { constexpr int j = J; fmac_bc<j, j == 0>(corr, grad, H[j]); });
    x -= corr;
    double s = bci0;
    static_for<6>([&](auto I) { constexpr int i = I; fmac_bc<i, i == 0>(s, x, a[i]); });
    const double sv = __shfl(s, myslot ? 8 + idk : 0, 16);
    const double rho = sel(myslot, -sv, 0.0);
    double dx = 0.0;
    static_for<6>([&](auto K) { constexpr int k = K; fmac_bc<k, k == 0>(dx, rho, NsT[k]); });
    x += dx;
  }
  QL_STAMP(26);
  x_out = x;
  return status;
}

// q (+) d = exp(d) * q as quat_box_plus (pose_core.hpp), with the Cody-Waite / minimax sincos of balance_core.hpp
// (one call for both, < 1 ulp) and a Newton reciprocal instead of two libm calls and a division
__device__ __forceinline__ void quat_box_plus_fast(const double q[4], const double d[3], double out[4]) {
  const double v2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  double e[4];
  if (v2 < 1e-24) {
    e[0] = 1.0; e[1] = 0.5 * d[0]; e[2] = 0.5 * d[1]; e[3] = 0.5 * d[2];
  } else {
    const double rv = rsqrt_nr(v2), v = v2 * rv;
    double sn, cs;
    sincos_reduced(0.5 * v, sn, cs);
    const double k = sn * rv;
    e[0] = cs; e[1] = k * d[0]; e[2] = k * d[1]; e[3] = k * d[2];
  }
  out[0] = e[0] * q[0] - e[1] * q[1] - e[2] * q[2] - e[3] * q[3];
  out[1] = e[0] * q[1] + e[1] * q[0] + e[2] * q[3] - e[3] * q[2];
  out[2] = e[0] * q[2] - e[1] * q[3] + e[2] * q[0] + e[3] * q[1];
  out[3] = e[0] * q[3] + e[1] * q[2] - e[2] * q[1] + e[3] * q[0];
}

// ---- linearisation + SQP loop -----------------------------------------------------------------------------
// pb: this problem's record (LDS, iteration order); pose: replicated, updated in place.
__device__ __forceinline__ int pose_sqp_coop(const PoseParamsDev &P, const PoseProblem &pb, bool live, double *lds_row,
                                             double pose[7], int &iters_out) {
  const int lr = threadIdx.x & 15;
  const int cj = lr - 8;
  // Support polygon (grid_map::Polygon: getCentroid and convertToInequalityConstraints, as polygon_centroid /
  // polygon_halfspaces of pose_core.hpp): the centroid replicated and branch-free (vertices beyond n_vertices contribute
  // nothing), and of the half-spaces only MY edge -- constraint lane 8 + e holds edge e (a degenerate edge is a constraint
  // that is not present; the reference compacts the rows, which keeps their order and changes nothing else).  Reciprocals
  // instead of the twelve divisions of the general form: the prologue of the kernel took 1.3 us with them.
  const int nv = pb.n_vertices;
  double centroid[2];
  {
    double sc = 0.0, cx = 0.0, cy = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool wrap = (i + 1 >= nv) || i == 3;
      const double nx = wrap ? pb.polygon[0][0] : pb.polygon[i < 3 ? i + 1 : 0][0];
      const double ny = wrap ? pb.polygon[0][1] : pb.polygon[i < 3 ? i + 1 : 0][1];
      const double cr = i < nv ? pb.polygon[i][0] * ny - nx * pb.polygon[i][1] : 0.0;
      sc += cr;
      cx = fma(cr, pb.polygon[i][0] + nx, cx);
      cy = fma(cr, pb.polygon[i][1] + ny, cy);
    }
    const double r3 = rcp_nr(3.0 * sc); // c / (6 area), area = sc / 2
    centroid[0] = cx * r3; centroid[1] = cy * r3;
  }
  double ga0 = 0.0, ga1 = 0.0, gbj = 0.0;
  bool edge_ok = false;
  {
    const int e = cj & 3;
    double mx = 0.0, my = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) { mx += i < nv ? pb.polygon[i][0] : 0.0; my += i < nv ? pb.polygon[i][1] : 0.0; }
    const double rn = rcp_nr((double)nv);
    mx *= rn; my *= rn;
    double px = 0.0, py = 0.0, qx = 0.0, qy = 0.0; // my edge: vertex e and its successor
#pragma unroll
    for (int i = 0; i < 4; i++) {
      px = sel(e == i, pb.polygon[i][0], px); py = sel(e == i, pb.polygon[i][1], py);
      const bool wrap = (i + 1 >= nv) || i == 3;
      qx = sel(e == i, wrap ? pb.polygon[0][0] : pb.polygon[i < 3 ? i + 1 : 0][0], qx);
      qy = sel(e == i, wrap ? pb.polygon[0][1] : pb.polygon[i < 3 ? i + 1 : 0][1], qy);
    }
    const double x1 = px - mx, y1 = py - my, x2 = qx - mx, y2 = qy - my;
    const double det = x1 * y2 - x2 * y1;
    edge_ok = e < nv && !(fabs(det) <= 1e-12 * (fabs(x1 * y2) + fabs(x2 * y1) + 1e-300));
    const double rd = rcp_nr(edge_ok ? det : 1.0);
    ga0 = (y2 - y1) * rd; ga1 = (x1 - x2) * rd;
    gbj = 1.0 + (ga0 * mx + ga1 * my);
  }
  // number of half-spaces of my problem: the valid edges on lanes 8..11 of my row
  const unsigned long long okm = __builtin_amdgcn_ballot_w64(edge_ok && lr >= 8 && lr < 12);
  const int nsp = __popc((unsigned)(okm >> (threadIdx.x & 48)) & 0xF00u);
  const int kleg = cj - 4;
  double lf[3] = {0, 0, 0}, lh[3] = {0, 0, 0}, lmax = 0.0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int c = 0; c < 3; c++) { lf[c] = sel(kleg == k, pb.stance[k][c], lf[c]); lh[c] = sel(kleg == k, pb.hips[k][c], lh[c]); }
    lmax = sel(kleg == k, pb.max_len[k], lmax);
  }
  const unsigned present = pb.present;
  const int nl = __popc(present & 0xFu);
  const bool cvalid = lr >= 8 && (cj < 4 ? edge_ok : ((present >> (kleg & 3)) & 1u) != 0);
  const int m = nsp + nl;

  // The objective's sums over the stance legs are linear in the rotation: with the per-problem constants
  //   Sd = sum d_k,  Sf = sum f_k,  M = sum d_k f_k'   (d_k the nominal stance in base coordinates, f_k the foothold)
  // and Pd_k = R d_k, e_k = p - f_k every one of them follows from R Sd (a rotation) and R M (a 3x3 product):
  //   sum Pd_k = R Sd                      sum (e_k + Pd_k) = nl p - Sf + R Sd
  //   sum Pd_k x e_k = (R Sd) x p - vee(R M)        (vee(A) = (A12 - A21, A20 - A02, A01 - A10))
  //   sum e_k.Pd_k = p.(R Sd) - trace(R M)        sum Pd_k e_k' = (R Sd) p' - R M
  // -- 36 multiply-adds per iteration where a loop over the legs takes four rotations, four cross products and 60 sums.
  double Sd[3] = {0, 0, 0}, Sf[3] = {0, 0, 0}, Mm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    const bool on = ((present >> kk) & 1u) != 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double d = on ? pb.nominal[kk][c] : 0.0, f = on ? pb.stance[kk][c] : 0.0;
      Sd[c] += d;
      Sf[c] += f;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double d = on ? pb.nominal[kk][c] : 0.0;
#pragma unroll
      for (int e = 0; e < 3; e++) Mm[c][e] = fma(d, pb.stance[kk][e], Mm[c][e]);
    }
  }
  const double w = P.com_weight;
  const double n0 = 2.0 * ((double)nl + w), n2 = 2.0 * (double)nl;
  double ut[3], ub[3], Gd[3]; // unit vector of my position row / rotation row (zero when I carry none); my part of the diagonal
#pragma unroll
  for (int j = 0; j < 3; j++) {
    ut[j] = lr == j ? 1.0 : 0.0;
    ub[j] = lr == 3 + j ? 1.0 : 0.0;
    Gd[j] = lr == j ? (j < 2 ? n0 : n2) : 0.0;
  }

  int k = 0, status = kStatusOk;
  bool sqp_done = !live;
  QL_STAMP(18);
  for (int outer = 0; outer < P.max_iter; outer++) {
    if (__all(sqp_done)) break;
    QL_STAMP(20);
    const double *p = pose;
    double R[9];
    quat_to_matrix(pose + 3, R);
    // ---- objective (replicated)
    double sPd[3], RM[3][3];
    rot(R, Sd, sPd);
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int e = 0; e < 3; e++) RM[i][e] = R[3 * i] * Mm[0][e] + R[3 * i + 1] * Mm[1][e] + R[3 * i + 2] * Mm[2][e];
    }
    const double fnl = (double)nl;
    const double sg[3] = {fnl * p[0] - Sf[0] + sPd[0], fnl * p[1] - Sf[1] + sPd[1], fnl * p[2] - Sf[2] + sPd[2]};
    double sx[3];
    cross3(sPd, p, sx);
    sx[0] -= RM[1][2] - RM[2][1]; sx[1] -= RM[2][0] - RM[0][2]; sx[2] -= RM[0][1] - RM[1][0];
    const double sdot = dot3(p, sPd) - (RM[0][0] + RM[1][1] + RM[2][2]);
    double sB[6];
    sB[0] = sPd[0] * p[0] - RM[0][0];
    sB[1] = 0.5 * ((sPd[0] * p[1] - RM[0][1]) + (sPd[1] * p[0] - RM[1][0]));
    sB[2] = 0.5 * ((sPd[0] * p[2] - RM[0][2]) + (sPd[2] * p[0] - RM[2][0]));
    sB[3] = sPd[1] * p[1] - RM[1][1];
    sB[4] = 0.5 * ((sPd[1] * p[2] - RM[1][2]) + (sPd[2] * p[1] - RM[2][1]));
    sB[5] = sPd[2] * p[2] - RM[2][2];
    // centre-of-mass term (weight w, planar): Pr = (R r_com)_xy, e = (p - centroid) with z = p_z
    double Pr3[3];
    rot(R, pb.r_com, Pr3);
    const double Pr[3] = {Pr3[0], Pr3[1], 0.0};
    const double pc[3] = {p[0] - centroid[0], p[1] - centroid[1], 0.0}; // pbar - rc
    const double ec[3] = {p[0] - centroid[0], p[1] - centroid[1], p[2]}; // p - rc (p enters skew(p) with its z)
    double cc[3];
    cross3(Pr, pc, cc);
    const double cdot = dot3(ec, Pr);
    // gradient (6) and Hessian: G = 2 [[nl I + w diag(1,1,0), -skew(q3)], [skew(q3), B]], q3 = sum Pd_k + w Pr
    const double g[6] = {sg[0] + w * (pc[0] + Pr[0]), sg[1] + w * (pc[1] + Pr[1]), sg[2], sx[0] + w * cc[0],
                         sx[1] + w * cc[1], sx[2] + w * cc[2]};
    const double qx = 2.0 * (sPd[0] + w * Pr[0]), qy = 2.0 * (sPd[1] + w * Pr[1]), qz = 2.0 * sPd[2];
    double B2[6]; // 2 B, symmetric 3x3, packed 00 01 02 11 12 22
    B2[0] = 2.0 * (sB[0] - sdot + w * (Pr[0] * ec[0] - cdot));
    B2[1] = 2.0 * sB[1] + w * (Pr[0] * ec[1] + ec[0] * Pr[1]);
    B2[2] = 2.0 * sB[2] + w * (Pr[0] * ec[2] + ec[0] * Pr[2]);
    B2[3] = 2.0 * (sB[3] - sdot + w * (Pr[1] * ec[1] - cdot));
    B2[4] = 2.0 * sB[4] + w * (Pr[1] * ec[2] + ec[1] * Pr[2]);
    B2[5] = 2.0 * (sB[5] - sdot + w * (Pr[2] * ec[2] - cdot));
    QL_STAMP(21);
    // my row of G: rows 0..2 = [diag(n0, n0, n2) | -S_c], rows 3..5 = [S_c | (2B)_c] with S_c = e_c x 2q the row c of
    // skew(2 q3) -- written with the lane's unit vectors (ut for a position row, ub for a rotation row, zero otherwise:
    // loop constants), so that the row is 27 multiply-adds instead of a pass of selects over the blocks
    double Gm[6], g0;
    {
      Gm[0] = Gd[0] + (ub[1] * qz - ub[2] * qy);
      Gm[1] = Gd[1] + (ub[2] * qx - ub[0] * qz);
      Gm[2] = Gd[2] + (ub[0] * qy - ub[1] * qx);
      Gm[3] = (qy * ut[2] - qz * ut[1]) + (ub[0] * B2[0] + ub[1] * B2[1] + ub[2] * B2[2]);
      Gm[4] = (qz * ut[0] - qx * ut[2]) + (ub[0] * B2[1] + ub[1] * B2[3] + ub[2] * B2[4]);
      Gm[5] = (qx * ut[1] - qy * ut[0]) + (ub[0] * B2[2] + ub[1] * B2[4] + ub[2] * B2[5]);
      g0 = 2.0 * ((ut[0] * g[0] + ut[1] * g[1] + ut[2] * g[2]) + (ub[0] * g[3] + ub[1] * g[4] + ub[2] * g[5]));
    }
    // ---- constraints: my row of CI = -A', ci0 = max - value (PoseOptimizationFunctionConstraints.cpp:95-194)
    double a[6] = {0, 0, 0, 0, 0, 0}, bci0 = 0.0;
    {
      // support half-space j: A = [GA_j, 0 | GA_j skew(Pr3)], value GA_j (p + Pr3)_xy
      const double cw0 = p[0] + Pr3[0], cw1 = p[1] + Pr3[1];
      const double hs_b = gbj - (ga0 * cw0 + ga1 * cw1);
      // (G3 skew(Pr3))_c with G3 = (ga0, ga1, 0): column c of skew(Pr3) dotted with G3
      const double hs3 = ga1 * Pr3[2], hs4 = -ga0 * Pr3[2], hs5 = ga0 * Pr3[1] - ga1 * Pr3[0];
      const double hs_a[6] = {-ga0, -ga1, -0.0, hs3 * 1.0, hs4 * 1.0, hs5 * 1.0};
      // limb-length bound of my leg slot: value |R'(f - p) - hip| = |f - p - R hip| (a rotation keeps the length), and the
      // gradient runs along the same vector, ln = (p + R hip - f) / |.|: one rotation and one reciprocal square root
      double Ph[3];
      rot(R, lh, Ph);
      double ln[3] = {p[0] + Ph[0] - lf[0], p[1] + Ph[1] - lf[1], p[2] + Ph[2] - lf[2]};
      const double len2 = dot3(ln, ln);
      const double rl = rsqrt_nr(len2);
      const double len = len2 * rl;
      ln[0] *= rl; ln[1] *= rl; ln[2] *= rl;
      double lx[3];
      cross3(ln, Ph, lx); // ln' skew(Ph) = (Ph x ln)' ... see below
      // CI rows 3..5 of the reference: ln[0] Hs[j] + ln[1] Hs[3+j] + ln[2] Hs[6+j] with Hs = skew(Ph)
      //   = (skew(Ph)' ln)_j = -(Ph x ln)_j = (ln x Ph)_j
      const double ll_a[6] = {-ln[0], -ln[1], -ln[2], lx[0], lx[1], lx[2]};
      const bool hs = cj < 4;
#pragma unroll
      for (int i = 0; i < 6; i++) a[i] = sel(hs, hs_a[i], ll_a[i]);
      bci0 = sel(hs, hs_b, lmax - len);
      // (a lane without a valid constraint keeps whatever this computes: its key is masked by cvalid in the QP, its
      // normal is never selected, its slack never summed)
    }
    QL_STAMP(22);
    // ---- QP and the update
    double x;
    const int st = qp6_coop<true>(Gm, g0, a, bci0, cvalid, m, P.dummy_equality != 0, sqp_done, lds_row, x);
    QL_STAMP(27);
    double dp[6];
    static_for<6>([&](auto I) { constexpr int i = I; dp[i] = bc<i>(x); });
    if (!sqp_done) {
      k++;
      status = st;
      if (st != kStatusOk) {
        sqp_done = true;
      } else {
        pose[0] += dp[0]; pose[1] += dp[1]; pose[2] += dp[2];
        double qn[4];
        quat_box_plus_fast(pose + 3, dp + 3, qn);
        pose[3] = qn[0]; pose[4] = qn[1]; pose[5] = qn[2]; pose[6] = qn[3];
        const double nrm2 = dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2] + dp[3] * dp[3] + dp[4] * dp[4] + dp[5] * dp[5];
        if (nrm2 < P.tol * P.tol && P.tol > 0.0) sqp_done = true; // |dp| < tol, sequencequadraticproblemsolver.cpp:72-76
      }
    }
    QL_STAMP(28);
  }
  iters_out = k;
  return status;
}


// ---- PoseOptimizationQP::optimize in the row layout (PoseOptimizationQP.cpp:42-140) ---------------------------
// Position only: min sum |x + R d_i - f_i|^2 s.t. the support half-spaces on (x + R r_com)_xy, with the reference's
// all-zero equality column -- the n = 3 problem on the six variable lanes of qp6_coop (rows 3..5 identity), the
// half-spaces on constraint lanes 8..11.  pose: replicated, position replaced on success.
__device__ __forceinline__ int pose_qp_coop(const PoseParamsDev &P, const PoseProblem &pb, bool live, double *lds_row,
                                            double pose[7]) {
  const int lr = threadIdx.x & 15, cj = lr - 8;
  double R[9], qv[3] = {0.0, 0.0, 0.0};
  quat_to_matrix(pose + 3, R);
  const unsigned present = pb.present;
  const int nl = __popc(present & 0xFu);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (!((present >> k) & 1u)) continue;
    double Rd[3];
    rot(R, pb.nominal[k], Rd);
#pragma unroll
    for (int i = 0; i < 3; i++) qv[i] += -2.0 * (pb.stance[k][i] - Rd[i]);
  }
  double GA[4][2], gb[4], Rr[3];
  const int m = polygon_halfspaces(pb.n_vertices, pb.polygon, GA, gb);
  rot(R, pb.r_com, Rr);
  double Gm[6], a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, g0 = 0.0, bci0 = 0.0;
#pragma unroll
  for (int j = 0; j < 6; j++) Gm[j] = lr == j ? (j < 3 ? 2.0 * (double)nl : 1.0) : 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++) g0 = sel(lr == i, qv[i], g0);
  const bool cvalid = lr >= 8 && cj < m;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const bool me = cvalid && cj == j;
    a[0] = sel(me, -GA[j][0], a[0]);
    a[1] = sel(me, -GA[j][1], a[1]);
    bci0 = sel(me, gb[j] - (GA[j][0] * Rr[0] + GA[j][1] * Rr[1]), bci0);
  }
  double x;
  const int st = qp6_coop<true>(Gm, g0, a, bci0, cvalid, m, P.dummy_equality != 0, !live, lds_row, x, 3);
  const double x0 = bc<0>(x), x1 = bc<1>(x), x2 = bc<2>(x);
  if (st == kStatusOk) { pose[0] = x0; pose[1] = x1; pose[2] = x2; }
  return st;
}

// ---- PoseOptimizationGeometric::optimize in the row layout (PoseOptimizationGeometric.cpp:34-105) --------------
// Everything but the eigen-problem is a few dozen replicated operations (pose_core.hpp: pose_geometric).  The 4x4
// symmetric eigen-problem runs on the four lanes of a quad, lane i holding row i of A and of the vectors V: the
// parallel Jacobi ordering (0,1)(2,3) / (0,2)(1,3) / (0,3)(1,2) rotates two disjoint pairs at once, so a sweep is
// three rounds instead of six rotations; partners and the other pair's (c, s) arrive through quad_perm.
template <int kPartner> // quad_perm control that maps lane i to its partner in this round
__device__ __forceinline__ void jacobi_round(double (&A)[4], double (&V)[4], int c4, bool &rotated) {
  // this round's pairs: partner = c4 ^ mask with mask 1, 2, 3 for kPartner 0xB1 [1,0,3,2], 0x4E [2,3,0,1], 0x1B [3,2,1,0]
  constexpr int mask = kPartner == 0xB1 ? 1 : kPartner == 0x4E ? 2 : 3;
  const int pr = c4 ^ mask;
  const bool low = c4 < pr;                 // I am p (the smaller index) of my pair
  // a_pp, a_qq, a_pq of my pair: my diagonal, my partner's diagonal, my element in my partner's column
  double dme = 0.0, ape = 0.0;
#pragma unroll
  for (int k = 0; k < 4; k++) { dme = sel(c4 == k, A[k], dme); ape = sel(pr == k, A[k], ape); }
  const double dpa = dpp<kPartner>(dme), apa = dpp<kPartner>(ape);
  const double app = low ? dme : dpa, aqq = low ? dpa : dme, apq = low ? ape : apa; // both lanes use A[p][q] of lane p
  // rotation (Rutishauser): t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq).  An
  // off-diagonal element that no longer registers against both diagonal elements (100 |apq| below their last bit) is
  // set to zero instead of being rotated away: the sweeps then end by themselves, and theta cannot overflow.
  const double g100 = 100.0 * fabs(apq);
  const bool live = apq != 0.0 && !((fabs(app) + g100 == fabs(app)) && (fabs(aqq) + g100 == fabs(aqq)));
  rotated = rotated || live;
  const double theta = (aqq - app) * (0.5 * rcp_nr(live ? apq : 1.0));
  const double th1 = theta * theta + 1.0;
  const double t = (theta >= 0.0 ? 1.0 : -1.0) * rcp_nr(fabs(theta) + th1 * rsqrt_nr(th1));
  const double cr = live ? rsqrt_nr(t * t + 1.0) : 1.0, sr = live ? t * cr : 0.0;
  // (c, s) of the pair that owns column k, for every k: mine for my two columns, the other pair's for the others
  constexpr int kOther = mask == 1 ? 0x4E : 0xB1; // quad_perm reaching a lane of the other pair
  const double co = dpp<kOther>(cr), so = dpp<kOther>(sr);
  // columns: A <- A J, V <- V J with J = product of the two rotations; column pairs (p, q) are compile-time per round
  const auto cols = [&](double (&M)[4]) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int q = p ^ mask;
      if (p > q) continue;
      // the pair (p, q): its rotation is mine if p or q is my index or my partner's, else the other pair's
      const bool minep = (p == c4) || (q == c4);
      const double c = minep ? cr : co, sn = minep ? sr : so;
      const double mp = M[p], mq = M[q];
      M[p] = c * mp - sn * mq;
      M[q] = sn * mp + c * mq;
    }
  };
  cols(A);
  cols(V);
  // rows: A <- J' A: my row mixes with my partner's; the element the rotation annihilates is exactly zero afterwards
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double other_row = dpp<kPartner>(A[k]);
    const double mixed = low ? cr * A[k] - sr * other_row : sr * other_row + cr * A[k];
    A[k] = pr == k ? 0.0 : mixed;
  }
}

// C: row c4 of the symmetric matrix on every lane of the quad.  Returns component c4 of the eigenvector of the
// largest eigenvalue, as sym4_eigen + the selection in pose_geometric do.
__device__ __forceinline__ double sym4_dominant_quad(const double (&Crow)[4], int c4) {
  double A[4], V[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { A[k] = Crow[k]; V[k] = c4 == k ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 12; sweep++) { // quadratic convergence: 4-6 sweeps, ended by a sweep without a rotation
    bool rotated = false;
    jacobi_round<0xB1>(A, V, c4, rotated);
    jacobi_round<0x4E>(A, V, c4, rotated);
    jacobi_round<0x1B>(A, V, c4, rotated);
    if (__builtin_amdgcn_ballot_w64(rotated) == 0ull) break;
  }
  // eigenvalue i = A[i][i] on lane i; V[k] on lane r = component r of eigenvector k
  double w = 0.0;
#pragma unroll
  for (int k = 0; k < 4; k++) w = sel(c4 == k, A[k], w);
  const double w0 = quad_bc<0>(w), w1 = quad_bc<1>(w), w2 = quad_bc<2>(w), w3 = quad_bc<3>(w);
  double wb = w0, comp = V[0];
  if (w1 > wb) { wb = w1; comp = V[1]; }
  if (w2 > wb) { wb = w2; comp = V[2]; }
  if (w3 > wb) { wb = w3; comp = V[3]; }
  return comp;
}

// pose: replicated on the row (output).  sfo: stance for orientation by limb id, replicated.
__device__ __forceinline__ void pose_geometric_coop(const PoseProblem &pb, const double sfo[4][3], double pose[7]) {
  const int c4 = threadIdx.x & 3;
  double cen[2];
  polygon_centroid(pb.n_vertices, pb.polygon, cen);
  // my row of C = sum_k Ak'Ak ... as pose_geometric, only row c4
  double z = 0.0, Crow[4] = {0.0, 0.0, 0.0, 0.0}, Am[16];
#pragma unroll
  for (int i = 0; i < 16; i++) Am[i] = 0.0;
  int nl = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (!((pb.present >> k) & 1u)) continue;
    nl++;
    z += pb.stance[k][2] - pb.nominal[k][2];
    const double *a = pb.stance[k], *b = pb.nominal[k];
    const double Ak[16] = {0.0,         -a[0] + b[0], -a[1] + b[1], -a[2] + b[2],
                           a[0] - b[0], 0.0,          -a[2] - b[2], a[1] + b[1],
                           a[1] - b[1], a[2] + b[2],  0.0,          -a[0] - b[0],
                           a[2] - b[2], -a[1] - b[1], a[0] + b[0],  0.0};
    double arow[4] = {0.0, 0.0, 0.0, 0.0}; // row c4 of Ak
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) arow[j] = sel(c4 == i, Ak[4 * i + j], arow[j]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < 4; m++) acc += arow[m] * Ak[4 * m + j];
      Crow[j] += acc;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) Am[i] += Ak[i];
  }
  const double n = (double)nl, rn = rcp_nr(n);
  z *= rn;
#pragma unroll
  for (int i = 0; i < 16; i++) Am[i] *= rn;
  {
    double arow[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) arow[j] = sel(c4 == i, Am[4 * i + j], arow[j]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < 4; m++) acc += arow[m] * Am[4 * m + j];
      Crow[j] -= n * acc;
    }
  }
  // (C is symmetric to the last bit: Ak is antisymmetric by construction, so (Ak Ak)[i][j] and [j][i] are the same
  // products summed in the same order; sym4_eigen's 1/2 (C + C') is the identity on it)
  const double qc = sym4_dominant_quad(Crow, c4);
  double q[4] = {quad_bc<0>(qc), quad_bc<1>(qc), quad_bc<2>(qc), quad_bc<3>(qc)};
  const double inq = rsqrt_nr(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] *= inq;
  { // setUnique: first non-zero component positive
    bool neg = false, decided = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool here = !decided && q[i] != 0.0;
      neg = here ? q[i] < 0.0 : neg;
      decided = decided || here;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = neg ? -q[i] : q[i];
  }
  pose_geometric_finish(pb, sfo, cen, z, q, pose);
}

// ---- BaseAuto::optimizePose in the row layout (BaseAuto.cpp:394-400): geometric -> QP -> check -> SQP ------------
__device__ __forceinline__ int base_auto_coop(const PoseParamsDev &P, const PoseProblem &pb, const double sfo[4][3],
                                              const double min_len[4], double leg_tol, bool live, double *lds_row,
                                              double pose[7], int &stage, int &iters) {
  pose_geometric_coop(pb, sfo, pose);
  stage = 2;
  iters = 0;
  int st = pose_qp_coop(P, pb, live, lds_row, pose);
  const bool need = live && st == kStatusOk && !pose_check(pb, pose, min_len, leg_tol);
  if (__builtin_amdgcn_ballot_w64(need) != 0ull) { // the SQP only for the rows the checker rejects
    int it = 0;
    const int s2 = pose_sqp_coop(P, pb, need, lds_row, pose, it);
    if (need) { st = s2; stage = 3; iters = it; }
  }
  return st;
}

} // namespace coop
} // namespace qlamd
