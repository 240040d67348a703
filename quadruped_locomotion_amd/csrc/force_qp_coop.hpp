// The contact-force QP of one robot on its 16 lanes: inverse of G, unconstrained minimiser, the active-set loop and
// the refinement on the final working set -- the part of the balance step (balance_coop.hpp) and of the whole-body step
// (wholebody_coop.hpp) that is the same.  Device-only (gfx950).
//
//   min 1/2 x'G x + g0'x   over the 12 contact-force components (lane 4 leg + c holds component c of leg `leg` and row
//   3 leg + c of every matrix; c = 3 is a spare lane; rows of a swing leg are padding: unit diagonal, g0 = 0)
//   s.t. per stance leg   n'f >= f_min,   mu n'f +- t1'f >= 0,   mu n'f +- t2'f >= 0          (kinds 0, 1..4)
//   kTorque:              |tau0_k - J[:,k]'f| <= tau_max for the leg's joints k = 0..2           (kinds 5+2k upper, 6+2k lower)
// Every row touches the three variables of ONE leg, which is what the loop is built on: lane (leg, c) watches friction
// row c + 1 of its leg, the minimum-force row when c = 0 and, with kTorque, the two torque bounds of joint c.
//
// Method: Goldfarb-Idnani with the operators of the original paper kept explicitly (see balance_coop.hpp's header),
// pivot rule, step lengths and termination test of QuadProg++ (QuadProg++.cc:216-445).
#pragma once

#include "coop_lanes.hpp"

// The ghost rows of the active-set loop (below) lean on IEEE semantics: a finished row's candidate normal is zero, so its
// step length is infinite, its first ghost step turns its x into NaN (inf * 0), every comparison on its slacks is false from
// then on and its selection key is 0 -- which points at the table entry its latch zeroed.  Under finite-math assumptions
// the compiler may fold those comparisons and a ghost row whose slacks are still slightly violated (one that stopped on
// the feasibility tolerance, QuadProg++.cc:246) would pick a real normal and disturb the H and N* its refinement reads.
#if defined(__FAST_MATH__) || (defined(__FINITE_MATH_ONLY__) && __FINITE_MATH_ONLY__)
#error "force_qp_coop.hpp needs IEEE inf / NaN semantics: do not build this translation unit with -ffast-math / -ffinite-math-only"
#endif

// the warm block's two workgroup stamps (tools/stamp_probe_warm_loop.py) belong to the balance kernel's unit; a unit with its
// own use of slots 1 and 2 (the tick) defines QL_QP_BLOCK_STAMP away before including this file
#ifndef QL_QP_BLOCK_STAMP
#define QL_QP_BLOCK_STAMP(slot) QL_BLOCK_STAMP(slot)
#endif
namespace qlamd {
namespace coop {

// LDS of one robot: 144 doubles.  After the loop: the N* export of the refinement.  During the loop: what a finished row
// latches (slots 0..47), the row a drop exports (kDropSlot..+11) and a zero for lanes that have no component of it to read.
// (144 and not more: with the model table and the normals table a wavefront then needs 13 056 bytes of LDS, and twelve
// wavefronts -- three per SIMD -- fit a compute unit's 160 KB.)
constexpr int kCoopLdsDoubles = 12 * 12;
constexpr int kMaxRefinePasses = 8; // of a warm start whose set did not fit (force_qp_coop's refinement)
#ifndef QLAMD_REFINE_AGAIN
#define QLAMD_REFINE_AGAIN 1e-3
#endif
constexpr double kRefineAgain = QLAMD_REFINE_AGAIN; // a warm start's refinement passes again while a pass moved x by more than this
constexpr int kDropSlot = 48, kZeroSlot = 60, kWarmSlot = 61; // (kWarmSlot: the update count of a warm start, an int)
// rows of the wavefront's table of constraint normals ([row kind][lane]) the QP uses: 5, and 3 more with kTorque
constexpr int kForceQpNrmRows = 5, kForceQpNrmRowsTorque = 8;
template <class T>
constexpr T one_v = T(1);
constexpr int kStatusWarmRejected = 6; // QLAMD_STATUS_WARM_REJECTED: a warm start whose answer did not check out
struct ForceQp {
  double Gm[12];                 // my row of G
  double g0;                     // my entry of g0
  double nb[3], t1[3], t2[3];    // my leg's contact normal and tangents (base frame), whole vectors
  double myn, myt1, myt2;        // their component c (0 on the spare lane)
  double mu, f_min;
  bool on, comp;                 // my leg supports; I carry a variable (c < 3)
  int nS;                        // number of stance legs
  int refine_passes;
  // kTorque only.  tau = tau0 - J_leg' f:  J[c][k] and J[k][c] of my leg (0 unless on && comp)
  double jrow[3], jcol[3];
  double tq_up, tq_lo;           // tau_max - tau0_c, tau_max + tau0_c
  // kWarm only: the working set to start from (bit kKinds leg + kind, kKinds = 5 or, with kTorque, 11; as written to `ws_out`
  // by an earlier solve) and the stance legs as a bit mask (rows of other legs are left out of it)
  unsigned long long warm;
  unsigned stance;
  bool build_set;                // kWarm: a robot without a set builds one by rounds (a caller that hands in sets); false: the
                                 // method of the reference from the empty set, bit for bit the cold kernels' answer
};

// My row of G and my entry of g0 for the objective  |A f - F|^2_S + w_reg |f|^2  (A = [1 ... ; [r_leg]x ...], the
// distribution of the target wrench F = [force ; moment] over the contact forces, ContactForceDistribution.cpp:184-252),
// plus, when `jrow` is given, the torque term  w_tau |tau0 - J' f|^2  of the whole-body step on my own leg's block.
// foot: component c of my leg's foot position; stance: bit per leg; row_on: my leg supports and I carry a variable.
// S: the six weights; jrow[k] = J[c][k] of my leg, jt0 = sum_k J[c][k] tau0_k.
__device__ __forceinline__ void force_qp_objective(const double S[6], double w_reg, double foot, unsigned stance, bool row_on,
                                                   const double F[6], const double *jrow, double w_tau_jt0, double Gm[12],
                                                   double &g0, double w_tau = 0.0) {
  const int lr = threadIdx.x & 15, leg = lr >> 2, c = lr & 3;
  const int myidx = 3 * leg + c;
  // foot positions of all legs, replicated
  double r[4][3];
  static_for<12>([&](auto J) { constexpr int j = J; r[j / 3][j % 3] = bcv<j>(foot); });
  const double rl[3] = {quad_bc<0>(foot), quad_bc<1>(foot), quad_bc<2>(foot)};
  // a = r_leg x e_c  (column c of skew(r_leg))
  const double a[3] = {sel(c == 1, -rl[2], sel(c == 2, rl[1], 0.0)), sel(c == 0, rl[2], sel(c == 2, -rl[0], 0.0)),
                       sel(c == 0, -rl[1], sel(c == 1, rl[0], 0.0))};
  const double sa[3] = {S[3] * a[0], S[4] * a[1], S[5] * a[2]};
  const double Sfc = pick3(S, c);
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const bool both = row_on && ((stance >> m) & 1u);
    const double xp = r[m][0], yp = r[m][1], zp = r[m][2];
    // (r_m x e_b) for b = 0,1,2: (0,z',-y'), (-z',0,x'), (y',-x',0)
    const double e0 = sa[1] * zp - sa[2] * yp;
    const double e1 = -sa[0] * zp + sa[2] * xp;
    const double e2 = sa[0] * yp - sa[1] * xp;
    Gm[3 * m + 0] = both ? e0 + (c == 0 ? Sfc : 0.0) : 0.0;
    Gm[3 * m + 1] = both ? e1 + (c == 1 ? Sfc : 0.0) : 0.0;
    Gm[3 * m + 2] = both ? e2 + (c == 2 ? Sfc : 0.0) : 0.0;
  }
  if (jrow) { // w_tau J J' on my own leg's block: entry (c, b) = sum_k J[c][k] J[b][k]
    double jj[3];
    static_for<3>([&](auto Bq) {
      constexpr int bq = Bq;
      jj[bq] = jrow[0] * quad_bc<bq>(jrow[0]) + jrow[1] * quad_bc<bq>(jrow[1]) + jrow[2] * quad_bc<bq>(jrow[2]);
    });
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const bool own = row_on && m == leg;
#pragma unroll
      for (int bq = 0; bq < 3; bq++) Gm[3 * m + bq] += own ? w_tau * jj[bq] : 0.0;
    }
  }
  // rows of legs that do not support are padding: unit diagonal (they never meet a stance row)
#pragma unroll
  for (int j = 0; j < 12; j++)
    if (c < 3 && j == myidx) Gm[j] += row_on ? w_reg : 1.0;
  const double Fc = pick3(F, c);
  const double ST[3] = {S[3] * F[3], S[4] * F[4], S[5] * F[5]};
  const double g0v = -(Sfc * Fc + (a[0] * ST[0] + a[1] * ST[1] + a[2] * ST[2]) + w_tau_jt0);
  g0 = sel(row_on, g0v, 0.0);
}

// Returns the status; x: my component of the minimiser (valid for kStatusOk); iters_out: outer iterations taken
// (QuadProg++'s `iter`, QuadProg++.cc:216,262: one per candidate selected, the last one finds none) -- what a caller can
// hand back as a placement hint on the next control step (qlamd_placement_from_iterations).
// kWarm: the dual method starts from the working set Q.warm instead of the empty one -- the final working set of the same
// robot's previous control step, which at 400 Hz is this step's final set for 96 % of the robots of the bench batches
// (tools/experiments/warm_start_model.py) -- and the final working set goes to ws_out.  The rows are installed as equalities
// one after the other (directions, a full step onto the row, the rank-one update: no step lengths, no selection), then the
// slots whose multiplier came out negative are dropped, most negative first, until the pair is dual feasible; from there the
// method runs as always, so a set that no longer fits costs passes, not the answer (the minimiser is unique).  The passes a
// robot then still needs are its `iters_out`.  QuadProg++ has no such entry (solve_quadprog always starts from the
// unconstrained minimiser, QuadProg++.cc:216-233): iteration counts no longer match the reference's one for one, torques do.
template <bool kTorque, bool kWarm = false, int kLegs = 4, bool kRounds = true>
__device__ __forceinline__ int force_qp_coop(const ForceQp &Q, double *lds_row, double *lds_nrm, double &x, int &iters_out,
                                             unsigned long long *ws_out = nullptr) {
  using mask_t = std::conditional_t<kTorque, unsigned long long, unsigned>;
  constexpr int kKinds = kTorque ? 11 : 5;
  // kLegs: the legs that can support, in rows 0 .. kLegs-1 of the 16-lane row (the callers put the support legs first: a
  // robot on two legs is a 6-variable problem, and every product below is half as long); kV variables, at most kV slots
  static_assert(kLegs == 2 || kLegs == 4, "legs in front of the row");
  constexpr int kV = 3 * kLegs;
  const int lr = threadIdx.x & 15, leg = lr >> 2, c = lr & 3;
  const bool comp = Q.comp && leg < kLegs, on = Q.on, row_on = comp && on;
  const int myidx = 3 * leg + c, nS = Q.nS;
  const double eps = 2.220446049250313e-16;
  const double inf = INFINITY;
  const double mu = Q.mu, f_min = Q.f_min, g0 = Q.g0;
  const double myn = Q.myn, myt1 = Q.myt1, myt2 = Q.myt2;
  const double *nb = Q.nb, *t1 = Q.t1, *t2 = Q.t2, *Gm = Q.Gm;
  double H[kV];
  double c1 = 0.0, c2 = 0.0;
  {
#pragma unroll
    for (int j = 0; j < kV; j++) H[j] = Gm[j];
    // trace(G) over the stance block
    {
      double diag = 0.0;
#pragma unroll
      for (int j = 0; j < kV; j++) diag = (j == myidx) ? Gm[j] : diag;
      c1 = row_sum(sel(row_on, diag, 0.0));
    }
    // in-place Gauss-Jordan inversion, row per lane; pivot k = L_kk^2 of the Cholesky factor.
    // Row update H[j] -= f * H_k[j] is one v_fmac_f64_dpp (pivot row read through the DPP operand);
    // on the pivot lane f = 1 - 1/d turns the same formula into H_k[j] / d.
    bool bad = false;
    double my_pivot = 1.0; // pivot of my own row, for c2 below
    // The chain pivot -> reciprocal -> factor -> row updates -> next pivot is serial; the column of the NEXT pivot is
    // updated first, so that its reciprocal (hardware seed + one Newton step, 2e-15: the final refinement works on
    // G itself, not on this inverse) is under way while the other ten columns are still being updated.
    double d = bcv<0>(H[0]);
    static_for<kV>([&](auto K) {
      constexpr int k = K;
      bad = bad || !(d > 0.0);
      const double p = rcp_nr1(d);
      const bool piv = comp && (myidx == k);
      my_pivot = piv ? d : my_pivot;
      const double f = piv ? (1.0 - p) : H[k] * p;
      const double nf = -f;
      if constexpr (k < kV - 1) {
        fmac_bc<lane_of(k), true>(H[k + 1], H[k + 1], nf);
        d = bcv<k + 1>(H[k + 1]);
      }
      static_for<kV>([&](auto J) {
        constexpr int j = J;
        if constexpr (j != k && j != k + 1) fmac_bc<lane_of(k), (k == kV - 1 && j == 0)>(H[j], H[j], nf);
      });
      H[k] = piv ? p : nf;
    });
    // c2 = trace(J) = sum over the stance rows of 1/sqrt(pivot): one rsqrt per lane instead of one per pivot
    // (it only feeds the termination tolerance psi_tol)
    const double rp = rsqrt_nr(my_pivot);
    c2 = row_sum(sel(row_on, rp, 0.0));
    if (bad && nS > 0) { iters_out = 0; return kStatusNotPd; }
  }

  QL_STAMP(5);
  // ---------------------------------------------------------------- x0 = -H g0
  x = 0.0;
  {
    const double ng0 = -g0;
    double xa[3] = {0.0, 0.0, 0.0};
    static_for<kV>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(xa[j % 3], ng0, H[j]); });
    x = (xa[0] + xa[1]) + xa[2];
  }

  QL_STAMP(6); QL_QP_BLOCK_STAMP(6);
  // ---------------------------------------------------------------- active-set loop
  // One pass of the loop = one outer iteration of the dual method for every live row: zero or more drops of a blocking
  // constraint (each a rank-one update of H and N*), then the add of the candidate (another one), with the selection of the
  // next violated constraint (QuadProg++.cc:252-274) computed on the new x in the shadow of that last update: the two are
  // independent, so the broadcast chain of the update fills the wait states of the selection's cross-lane reduction and
  // vice versa.  Slots are NOT compacted on a drop: a freed slot lane is
  // reused by the next add (the order of the slots only breaks exact ties in the blocking-constraint search).
  //
  // A lone wavefront issues one instruction every ~4.5 cycles whatever its kind (tools/ubench/issue_model.hip), so a
  // pass costs what it has instructions.  The four robots of a wavefront take different branches of the method, which
  // would make every state update a predicated select; so the tail of a pass exists as two straight paths -- a row adds
  // (no predication, selection follows), a row drops (no selection) -- which the rows of a wavefront take one after the
  // other under their execution masks when they disagree, and as the general predicated form for what is left (a
  // degenerate add, an infeasible problem).  A row that has finished stays in the loop as a ghost (below).
  //
  // Selection.  Lane (leg, c) watches friction row c + 1 of its leg and, when c = 0, the minimum-force row: with the
  // three components of x_leg fetched through quad_perm each slack is a 3-term dot product with the lane's own row
  // vector.  The most violated row is found as the maximum of a 32-bit key per lane: the bits of the slack rounded to
  // single precision (negative floats order by magnitude as unsigned integers), low five bits replaced by lane and
  // row kind -- one v_max_u32 with a DPP operand per level instead of two moves and a v_min_f64.  Rows whose slacks
  // agree to 18 bits are ordered by lane; which of two almost equally violated rows enters first only changes the
  // path, the minimiser is unique.  Everything that decides a result in double precision stays in double precision:
  // whether a row is violated at all, the slack of the chosen row (fetched from its lane with ds_bpermute) and the
  // feasibility test |psi| <= tol (:246), which is only evaluated when the chosen slack is above -tol (psi <= the
  // most negative slack, so the test cannot pass otherwise).  Component c of the chosen row's normal comes from a
  // table in LDS ([row kind][lane], written once before the loop), read in the same shadow.
  double Ns[kV];
#pragma unroll
  for (int j = 0; j < kV; j++) Ns[j] = 0.0;
  double u = 0.0;            // multiplier of slot lr (free lanes: never read)
  int idk = 0;               // constraint id of slot lr
  unsigned used = 0;         // bit k set <=> slot lane k holds an active constraint
  int q = 0, iters = 0, status = kStatusOk;
  mask_t act_mask = 0, excl = 0;
  const double psi_tol = (double)(kKinds * nS) * eps * c1 * c2 * 100.0;
  double rnorm2 = 1.0; // R_norm^2
  // the rows that have finished, as a scalar mask: only ever assigned from a ballot taken where control is uniform (the
  // compiler has to know it for uniform, or every branch of the loop turns into execution-mask bookkeeping)
  unsigned long long done_m = 0ull;
  int ip = 0;
  double sp = 0.0, ucand = 0.0, npj = 0.0;
  // rows this lane evaluates
  const double fa = c == 0 ? 1.0 : c == 1 ? -1.0 : 0.0, fb = c == 2 ? 1.0 : c == 3 ? -1.0 : 0.0;
  const double Wf0 = mu * nb[0] + (fa * t1[0] + fb * t2[0]), Wf1 = mu * nb[1] + (fa * t1[1] + fb * t2[1]),
               Wf2 = mu * nb[2] + (fa * t1[2] + fb * t2[2]);
  const mask_t one = 1;
  const mask_t maskf = on ? (one << (kKinds * leg + c + 1)) : 0, maskm = (on && c == 0) ? (one << (kKinds * leg)) : 0;
  const mask_t masku = (kTorque && on && comp) ? (one << (kKinds * leg + 5 + 2 * c)) : 0, maskl = masku << 1;
  // key tag: lane, then what the lane's row is -- kTagBits low bits: 0 friction, 1 minimum force, 2 / 3 upper / lower torque bound
  constexpr int kTagBits = kTorque ? 2 : 1;
  constexpr unsigned kTagMask = (1u << (4 + kTagBits)) - 1u;
  const unsigned tagf = (unsigned)lr << kTagBits, tagm = tagf | 1u;
  const int row_addr = ((int)threadIdx.x & 48) << 2; // ds_bpermute byte address of lane 0 of my row
  const unsigned lanebit = 1u << lr;
  // table of normals: entry [kind][lane] = component c of my leg's row of that kind (0 minimum force, 1..4 friction)
  {
    const double nrm[5] = {myn, mu * myn + myt1, mu * myn - myt1, mu * myn + myt2, mu * myn - myt2};
#pragma unroll
    for (int k = 0; k < 5; k++) lds_nrm[64 * k + ((int)threadIdx.x & 63)] = nrm[k];
    lds_row[kZeroSlot] = 0.0;
    if constexpr (kTorque) {
#pragma unroll
      for (int k = 0; k < 3; k++) lds_nrm[64 * (5 + k) + ((int)threadIdx.x & 63)] = Q.jrow[k];
    }
  }
  // Ghost rows.  A row that has finished is not masked out of the loop: it runs along as a no-op.  Its
  // candidate normal is zero from then on (its entries of the normals table, which a finished row's keys always point at,
  // and its slots of the drop export are zeroed when it finishes), so z = r = n~ = 0 and both rank-one updates add exact
  // zeros to its H and N*; the two divisors of a pass are biased to 1 (zb: 1/16 per lane into the row sums of z'n_p and
  // n~'G n~); and what the epilogue needs of it (x, working set, status) is latched in its own LDS block at the moment it
  // finishes -- whatever else it computes afterwards (x, u, bookkeeping) is garbage nobody reads.  Why:
  // (1) EXEC stays full.  A lone wavefront issues the same instructions 7 % slower per pass when only one or two of its
  //     four 16-lane rows are enabled (tools/ubench/exec_mask_model.hip: dependent v_fma_f64 8.4 -> 10.6 cycles with two
  //     rows, independent ones 5.5 -> 6.7 with one; tools/experiments/row_mix_probe.py on this kernel), and the slowest
  //     robot of a small batch spends most of its passes as the only live row of its wavefront;
  // (2) every branch of the loop compares scalar masks: no execution-mask bookkeeping per pass.
  const int nt_slot = comp ? kDropSlot + myidx : kZeroSlot; // where a drop reads its component of n~ (a zero on spare lanes)
  double zb = 0.0;                              // 1/16 on ghost rows
  double s_up = 0.0, s_lo = 0.0; // kTorque: slacks of my joint's torque bounds, set by slacks()
  const auto slacks = [&](double xx, double &s_min, double &s_fric) {
    const double x0 = quad_bc<0>(xx), x1 = quad_bc<1>(xx), x2 = quad_bc<2>(xx);
    s_fric = fma(Wf2, x2, fma(Wf1, x1, Wf0 * x0));
    s_min = fma(nb[2], x2, fma(nb[1], x1, fma(nb[0], x0, -f_min)));
    if constexpr (kTorque) {
      const double d = fma(Q.jcol[2], x2, fma(Q.jcol[1], x1, Q.jcol[0] * x0));
      s_up = Q.tq_up + d;
      s_lo = Q.tq_lo - d;
    }
  };
  // sum of the violations of the rows this lane watches (QuadProg++.cc:238-245), after slacks()
  const auto violation = [&](double s_min, double s_fric) -> double {
    double v = vmin(0.0, s_fric) + sel(c == 0, vmin(0.0, s_min), 0.0);
    if constexpr (kTorque) v += sel(comp, vmin(0.0, s_up) + vmin(0.0, s_lo), 0.0);
    return v;
  };
  const auto umax_dpp = [](unsigned k, auto Ctrl) -> unsigned {
    constexpr int ctrl = decltype(Ctrl)::value;
    const unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, ctrl, 0xF, 0xF, true);
    return k > o ? k : o;
  };

  // rows that have just finished latch their results (slots 0..47 of their LDS block: free until the refinement) and
  // turn into ghosts
  const auto latch = [&](bool newly) {
    if (newly) {
      lds_row[lr] = x;
      reinterpret_cast<int2 *>(lds_row + 16)[lr] = make_int2((int)used, idk);
      reinterpret_cast<int2 *>(lds_row + 32)[lr] = make_int2(q | (iters << 8), status);
      lds_row[nt_slot] = 0.0;
      lds_nrm[64 * 1 + ((int)threadIdx.x & 63)] = 0.0; // a finished row's key is 0: lane 0, friction row 1
      zb = 0.0625; npj = 0.0;
    }
  };
  // Update of H and N* with the vectors of the step just taken (H[j] += hc * vec_j, N*[j] += nc * vec_j) and
  // selection of the next constraint at the new x, in one block so that the scheduler can weave the two (and the
  // bookkeeping of the step in front of them) together.  kMode 0: before the first step (no update; every live row
  // selects).  kMode 1: general -- rows in `resel` select (`fresh`: after an add, :252-262), the others keep their
  // candidate.  kMode 2: every live row has just added a constraint.
  double vec = 0.0, hc = 0.0, nc = 0.0;
  const auto update_and_select = [&](auto Mode, bool resel, bool fresh, bool finished = false) -> bool {
    constexpr int kMode = decltype(Mode)::value;
    constexpr bool kUpd = kMode != 0;
    if constexpr (kMode == 1) {
      iters += (resel && fresh) ? 1 : 0;
      excl = (resel && fresh) ? 0 : excl;
    } else {
      iters += 1;
      excl = 0;
    }
    const mask_t avail = ~(act_mask | excl);
    double s_min, s_fric;
    slacks(x, s_min, s_fric);
    unsigned kf = __float_as_uint((float)s_fric), km = __float_as_uint((float)s_min);
    kf = ((avail & maskf) != 0 && s_fric < 0.0) ? ((kf & ~kTagMask) | tagf) : 0u;
    km = ((avail & maskm) != 0 && s_min < 0.0) ? ((km & ~kTagMask) | tagm) : 0u;
    double myv = km > kf ? s_min : s_fric; // the slack behind this lane's key
    unsigned key = km > kf ? km : kf;
    if constexpr (kTorque) { // of the two bounds of a joint only the nearer one can be violated
      const bool lower = s_lo < s_up;
      const double s_t = lower ? s_lo : s_up;
      unsigned kt = __float_as_uint((float)s_t);
      kt = ((avail & (lower ? maskl : masku)) != 0 && s_t < 0.0) ? ((kt & ~kTagMask) | tagf | (lower ? 3u : 2u)) : 0u;
      myv = kt > key ? s_t : myv;
      key = kt > key ? kt : key;
    }
    if constexpr (kUpd) {
      static_for<3>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); });
    }
    key = umax_dpp(key, std::integral_constant<int, 0x128>{});
    if constexpr (kUpd) {
      static_for<3>([&](auto J) { constexpr int j = J + 3; fmac_bc<lane_of(j)>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); });
    }
    key = umax_dpp(key, std::integral_constant<int, 0x124>{});
    if constexpr (kUpd) {
      static_for<2>([&](auto J) { constexpr int j = J + 6; if constexpr (j < kV) { fmac_bc<lane_of(j)>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); } });
    }
    key = umax_dpp(key, std::integral_constant<int, 0x122>{});
    if constexpr (kUpd) {
      static_for<2>([&](auto J) { constexpr int j = J + 8; if constexpr (j < kV) { fmac_bc<lane_of(j)>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); } });
    }
    key = umax_dpp(key, std::integral_constant<int, 0x121>{});
    // the chosen row: lane and kind from the low bits, its slack from its lane, its normal from the table
    const int wl = (int)(key >> kTagBits) & 15;
    const int addr = row_addr + (wl << 2);
    const int vlo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(myv));
    const int vhi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(myv));
    int key_kind = (key & 1u) ? 0 : (wl & 3) + 1, tab_kind = key_kind;
    if constexpr (kTorque) {
      const bool tq = (key & 2u) != 0u;
      tab_kind = tq ? 5 + (wl & 3) : key_kind;
      key_kind = tq ? 5 + 2 * (wl & 3) + (int)(key & 1u) : key_kind;
    }
    const int key_ip = kKinds * (wl >> 2) + key_kind;
    double np_tab = lds_nrm[64 * tab_kind + ((int)threadIdx.x & 63)];
    if constexpr (kTorque) np_tab = ((key & 3u) == 3u) ? -np_tab : np_tab;
    if constexpr (kUpd) {
      static_for<2>([&](auto J) { constexpr int j = J + 10; if constexpr (j < kV) { fmac_bc<lane_of(j)>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); } });
    }
    const double np_new = sel((wl >> 2) == leg, np_tab, 0.0);
    const bool any = (int)key < 0;                       // a violated row that may enter
    const double v = __hiloint2double(vhi, vlo);
    // feasibility, QuadProg++.cc:246-250: only when the worst slack is within the tolerance can the sum be
    bool feasible = false;
    const bool close = (kMode != 1 || (resel && fresh)) && any && !(v < -psi_tol);
    if (__builtin_amdgcn_ballot_w64(close) != 0ull) {
      const double psi = (double)row_sum_f32((float)sel(on, violation(s_min, s_fric), 0.0));
      feasible = close && (fabs(psi) <= psi_tol);
    }
    const bool stop = !any || feasible || iters > kMaxOuter; // :271-274
    bool fin = stop; // this row finishes here
    if constexpr (kMode == 1) {
      status = (resel && stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      fin = (resel && stop) || finished;
      const bool take = resel && !stop;
      ip = take ? key_ip : ip;
      sp = sel(take, v, sp);
      ucand = sel(take, 0.0, ucand);
      npj = sel(take, np_new, npj);
    } else { // every row here selects: what a stopping row is left with is never read
      status = (stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      ip = key_ip; sp = v; ucand = 0.0; npj = np_new;
    }
    return fin;
  };
  // called where control is uniform, with the rows that have just finished (ghost rows report again: ignored)
  const auto note_finished = [&](bool fin) {
    const unsigned long long fin_m = __builtin_amdgcn_ballot_w64(fin);
    const unsigned long long newly_m = fin_m & ~done_m;
    done_m |= fin_m;
    if (newly_m != 0ull) latch(__builtin_amdgcn_inverse_ballot_w64(newly_m));
  };
  const auto update_only = [&]() {
    static_for<kV>([&](auto J) {
      constexpr int j = J;
      fmac_bc<lane_of(j), j == 0>(H[j], vec, hc);
      fmac_bc<lane_of(j)>(Ns[j], vec, nc);
    });
  };
  // dropping slot lpos (partial or dual-only step): n~ = row lpos of N* reaches the variable lanes through LDS,
  // then H += n~ n~'/e and N* -= (N* G n~) n~'/e with e = n~'G n~ (row lpos of N* becomes 0)
  double drop_einv = 0.0;
  const auto drop_vectors = [&](int lpos) {
    if (lr == lpos) {
#pragma unroll
      for (int j = 0; j < kV; j++) lds_row[kDropSlot + j] = Ns[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
    const double nt_me = lds_row[nt_slot];
    const int drop_id = __shfl(idk, lpos, 16);
    // (three partial sums each: a single accumulator makes twelve dependent broadcast-FMAs, 8.5 cycles apiece, of each product)
    double ga[3] = {0.0, 0.0, 0.0};
    static_for<kV>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(ga[j % 3], nt_me, Gm[j]); });
    const double Gn = (ga[0] + ga[1]) + ga[2];
    const double einv = rcp_nr1(row_sum(fma(nt_me, Gn, zb)));
    drop_einv = einv;
    double ca[3] = {0.0, 0.0, 0.0};
    static_for<kV>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(ca[j % 3], Gn, Ns[j]); });
    const double coef = (ca[0] + ca[1]) + ca[2];
    vec = nt_me;
    hc = nt_me * einv;
    // the dropped slot's own coefficient is n~'G n~ / e = 1: with exactly -1 its row N*[lpos][j] - n~_j is exactly 0
    // (n~ IS that row), which frees the slot without a pass of masked moves
    nc = sel(lr == lpos, -1.0, -coef * einv);
    return drop_id;
  };

  int warm_updates = 0; // rank-one updates of the warm start (kWarm)
  QL_QP_BLOCK_STAMP(1);
  if constexpr (kWarm) {
    // ---- warm start: install the previous working set (rows of legs that still support), then drop negative multipliers
    constexpr mask_t kLegRows = (one_v<mask_t> << kKinds) - 1;
    mask_t wm = (mask_t)Q.warm & (((Q.stance & 1u) ? kLegRows : 0) | ((Q.stance & 2u) ? kLegRows << kKinds : 0) |
                                  ((Q.stance & 4u) ? kLegRows << (2 * kKinds) : 0) | ((Q.stance & 8u) ? kLegRows << (3 * kKinds) : 0));
    // a working set holds at most three rows of a leg (every row touches the three variables of ONE leg: a fourth depends on
    // the others); a mask that claims more is not one, and the robot starts cold
    {
      bool sane = true;
#pragma unroll
      for (int l = 0; l < 4; l++) {
        const mask_t rows = (wm >> (kKinds * l)) & kLegRows;
        int n = 0;
        if constexpr (kTorque) n = __popcll((unsigned long long)rows);
        else n = __popc((unsigned)rows);
        sane = sane && n <= 3;
      }
      wm = sane ? wm : (mask_t)0;
    }
    // Which form: by rounds when some robot of the wavefront brings many rows (a round of four rows is 0.95 us whatever it
    // holds, a row on its own 0.40 us: the static bench batch -- 8 rows on average, up to 12 -- 15.5 -> 14.85 us by rounds, its
    // trot batches -- one to three rows on two legs -- 22.2 -> 23.2 us: profiles/r6/ab_install_forms.txt), row by row otherwise
    // and always in the throughput form of the kernels (!kRounds).  Wavefront-uniform: one scalar branch.
    // A robot WITHOUT a set -- its first step, or one whose support legs have just changed: the robots a launch of a trot waits
    // for -- builds one by rounds as well (kGreedy): in every round each leg takes its most violated row at the current x, the
    // round installs them as equalities, up to three rounds; negative multipliers are dropped below and the method of the
    // reference runs from there as after any warm start (same guarantees, same final check).  Four rows a round at 0.95 us
    // instead of one a pass at 0.65 us: a robot entering double support 9.5 passes -> three rounds and the passes that remain.
#ifdef QLAMD_NO_GREEDY
    constexpr bool kGreedy = false;   // (A/B builds)
#else
    constexpr bool kGreedy = kRounds;
#endif
    bool by_rounds = false, greedy = false;
    if constexpr (kRounds) {
      int nrows;
      if constexpr (kTorque) nrows = __popcll((unsigned long long)wm);
      else nrows = __popc((unsigned)wm);
      greedy = kGreedy && Q.build_set && wm == 0 && nS > 0;
      by_rounds = __builtin_amdgcn_ballot_w64(nrows >= (kLegs == 4 ? 6 : 4) || greedy) != 0ull;
    }
    if (by_rounds) {
      // Installed a ROUND at a time: the next row of every leg together.  Every row touches the variables of one leg, so the
      // directions of up to four rows -- z_m = H n_m, r_m = N* n_m -- are ONE broadcast sweep (column j of H and N* goes to the
      // accumulators of leg j / 3: 24 broadcast-FMAs for all of them, what a single row used to take), and the rows then go in one
      // after the other inside a straight block: d_k = n_k'z_k from a quad sum, the full step onto row k, the rank-one update, and the
      // directions of the legs still to come corrected by one broadcast-FMA each (z_m -= z_k (n_m'z_k) / d_k: the quad sum that gave
      // d_k on leg k's lanes gave n_m'z_k on leg m's) instead of being swept again.  A lone wavefront pays for dependent chains:
      // round 5 installed a row in 0.40 us (LDS fetch of the normal -> sweep -> two row sums -> reciprocal -> update, 960 cycles:
      // profiles/r6/warm_install_probe.txt), twelve rows in 4.8 us; a round of four is 0.7 us and three rounds are the most there are.
      // A leg without a row in a round rides along with a zero normal and divisors biased to 1, like a ghost row of the loop.
      unsigned mine = (unsigned)((wm >> (kKinds * leg)) & kLegRows); // the rows of my leg still to install (lanes of a leg agree)
      const bool any_greedy = kGreedy && __builtin_amdgcn_ballot_w64(greedy) != 0ull; // (scalar)
      // (the loop in two copies -- wavefronts with a robot that builds its set, and the others, whose rounds stay the straight
      // code they were: folded into one loop the selection costs the static bench batch 0.25 us a step)
      const auto rounds = [&](auto WithGreedy) {
      constexpr bool kWithGreedy = decltype(WithGreedy)::value;
      for (int round = 0;; round++) {
        bool have = mine != 0u;
        int kind = have ? __ffs((int)mine) - 1 : 0;
        mine &= mine - 1u;
        if constexpr (kWithGreedy) {
          // a robot that builds its set: the most violated row of my leg that is not in the set yet, by the keys of the
          // selection (update_and_select) reduced over the quad instead of the row
          if (round < 3) {
            double s_min, s_fric, xg = x;
            asm volatile("" : "+v"(xg)); // (not to be computed ahead of the branch, on the path of the wavefronts that build nothing)
            slacks(xg, s_min, s_fric);
            const mask_t avail = ~act_mask;
            unsigned kf = __float_as_uint((float)s_fric), km = __float_as_uint((float)s_min);
            kf = ((avail & maskf) != 0 && s_fric < 0.0) ? ((kf & ~kTagMask) | tagf) : 0u;
            km = ((avail & maskm) != 0 && s_min < 0.0) ? ((km & ~kTagMask) | tagm) : 0u;
            unsigned key = km > kf ? km : kf;
            if constexpr (kTorque) { // of the two bounds of a joint only the nearer one can be violated
              const bool lower = s_lo < s_up;
              const double s_t = lower ? s_lo : s_up;
              unsigned kt = __float_as_uint((float)s_t);
              kt = ((avail & (lower ? maskl : masku)) != 0 && s_t < 0.0) ? ((kt & ~kTagMask) | tagf | (lower ? 3u : 2u)) : 0u;
              key = kt > key ? kt : key;
            }
            key = umax_dpp(key, std::integral_constant<int, 0xB1>{}); // quad_perm [1,0,3,2]
            key = umax_dpp(key, std::integral_constant<int, 0x4E>{}); // quad_perm [2,3,0,1]
            const bool ghave = greedy && (int)key < 0;
            const int cw = (int)((key >> kTagBits) & 3u); // the lane of my quad that watches the row
            int gkind = (key & 1u) ? 0 : cw + 1;
            if constexpr (kTorque) gkind = (key & 2u) ? 5 + 2 * cw + (int)(key & 1u) : gkind;
            have = greedy ? ghave : have;
            kind = greedy ? gkind : kind;
          }
        }
        if (__builtin_amdgcn_ballot_w64(have) == 0ull) break;
        // component c of my leg's row and its offset: n'x - b >= 0 with b = f_min (kind 0), 0 (friction), and for the torque bounds
        // of joint k (kinds 5 + 2k upper, 6 + 2k lower) n = +-J[:,k], b = -(tau_max -+ tau0_k) -- held by the joint's lane
        double nv, bp;
        {
          const double fr = sel(kind == 1, myt1, sel(kind == 2, -myt1, sel(kind == 3, myt2, -myt2)));
          nv = sel(kind == 0, myn, fma(mu, myn, fr));
          bp = sel(kind == 0, f_min, 0.0);
          if constexpr (kTorque) {
            const bool tq = kind >= 5, lower = tq && ((kind - 5) & 1) != 0;
            const int kj = (kind - 5) >> 1;
            const double jn = pick3(Q.jrow, kj);
            nv = sel(tq, sel(lower, -jn, jn), nv);
            const double bu = sel(kj == 0, quad_bc<0>(Q.tq_up), sel(kj == 1, quad_bc<1>(Q.tq_up), quad_bc<2>(Q.tq_up)));
            const double bl = sel(kj == 0, quad_bc<0>(Q.tq_lo), sel(kj == 1, quad_bc<1>(Q.tq_lo), quad_bc<2>(Q.tq_lo)));
            bp = sel(tq, -sel(lower, bl, bu), bp);
          }
        }
        npj = sel(have && comp, nv, 0.0);
        // which row every leg brings, for all lanes of the robot: (kind + 1) in four bits per leg, OR-ed over the row
        unsigned kinds = (have && c == 0) ? (unsigned)(kind + 1) << (4 * leg) : 0u;
        static_for<4>([&](auto K) {
          constexpr int ctrl = K == 0 ? 0x128 : K == 1 ? 0x124 : K == 2 ? 0x122 : 0x121;
          kinds |= (unsigned)__builtin_amdgcn_mov_dpp((int)kinds, ctrl, 0xF, 0xF, true);
        });
        double sl = quad_sum(npj * x) - sel(have, bp, 0.0); // slack of my leg's row at x, kept up to date through the round
        double za[kLegs], ra[kLegs];
#pragma unroll
        for (int m = 0; m < kLegs; m++) { za[m] = 0.0; ra[m] = 0.0; }
        static_for<kV>([&](auto J) {
          constexpr int j = J;
          fmac_bc<lane_of(j), j == 0>(za[j / 3], npj, H[j]);
          fmac_bc<lane_of(j)>(ra[j / 3], npj, Ns[j]);
        });
        static_for<kLegs>([&](auto K) {
          constexpr int k = K;
          // (a lone wavefront pays per instruction, and a round is four of these: everything a leg without a row, or with a row
          // that depends on the rows before it, must not do is done by ONE factor okf = 0 on its directions -- its z, r and
          // step are then exact zeros all the way down -- instead of a select per quantity)
          const int kk = (int)((kinds >> (4 * k)) & 15u) - 1; // leg k's row of this round (-1: none)
          const double qk = quad_sum(npj * za[k]);            // on the lanes of leg m: n_m'z_k
          const double dk = bc<4 * k>(qk);
          // a row that depends on the rows installed before it is left out: z'n_p is then rounding noise, which with the entries
          // of H reaching 1 / w_reg = 1e4 means up to 1e-10, while an independent row has z'n_p >= |n|^2 / trace(G) ~ 1e-3
          const bool ok = kk >= 0 && dk > 1e-6;
          const double okf = sel(ok, 1.0, 0.0);
          const double zi = rcp_nr1(sel(ok, dk, 1.0)) * okf;  // 1 / d_k, or 0
          const double zk = za[k] * okf, rr = ra[k] * okf;
          const double tw = -bc<4 * k>(sl) * zi;
          x = fma(tw, zk, x);
          u = fma(-tw, rr, u);
          sl = fma(tw, qk, sl);
          const int newlane = __ffs(~used & 0xFFFu) - 1;
          const bool newslot = ok && lr == newlane;
          const int pk = kKinds * k + kk;
          vec = zk * zi;
          hc = -zk;
          const double rk = sel(newslot, -1.0, rr);
          nc = -rk;
          u = sel(newslot, tw, u);
          idk = newslot ? pk : idk;
          used |= ok ? (1u << newlane) : 0u;
          act_mask |= ok ? (one << pk) : 0;
          rnorm2 = vmax(rnorm2, dk); // (a row left out has d_k <= 1e-6 < R_norm^2)
          q += ok ? 1 : 0;
          // the legs still to come: z_m -= z_k (n_m'z_k) / d_k, r_m -= r_k (n_m'z_k) / d_k (the new slot's own entry: + n_m'z_k / d_k)
          if constexpr (k + 1 < kLegs) {
            const double cm = -qk * zi;
            static_for<kLegs - 1 - k>([&](auto M) {
              constexpr int m = k + 1 + M;
              fmac_bc<4 * m, M == 0>(za[m], cm, zk);
              fmac_bc<4 * m>(ra[m], cm, rk);
            });
          }
          update_only();
        });
      }
      };
      if constexpr (kGreedy) {
        if (__builtin_expect(any_greedy, 0)) rounds(std::true_type{}); // (out of line: the common path stays contiguous)
        else rounds(std::false_type{});
      } else {
        rounds(std::false_type{});
      }
    } else {
      // One row after the other (the throughput form of the kernels, which has no registers for a round's eight direction
      // vectors: in 168 registers the round form spills 70 values around the loop, 65 536 warm-started trot robots 70 -> 85 us):
      // directions as in a pass, a full step onto the row, the rank-one update.
      for (;;) {
        const bool has = wm != 0;
        if (__builtin_amdgcn_ballot_w64(has) == 0ull) break;
        int p = 0;
        if constexpr (kTorque) p = has ? __ffsll((long long)wm) - 1 : 0;
        else p = has ? __ffs((int)wm) - 1 : 0;
        wm &= wm - 1;
        const int pleg = kTorque ? ((p * 47) >> 9) : id_leg(p), kind = p - kKinds * pleg;
        // the row's normal on my lane and its offset: n'x - b >= 0 with b = f_min (kind 0), 0 (friction), and for the torque
        // bounds of joint k (kinds 5 + 2k upper, 6 + 2k lower) n = +-J[:,k], b = -(tau_max -+ tau0_k) -- held by the joint's lane
        int tab_kind = kind;
        double sign = 1.0, b_p = sel(kind == 0, f_min, 0.0);
        if constexpr (kTorque) {
          const int kj = (kind - 5) >> 1;
          const bool tq = kind >= 5, lower = tq && ((kind - 5) & 1) != 0;
          tab_kind = tq ? 5 + kj : kind;
          sign = lower ? -1.0 : 1.0;
          const int src = 4 * pleg + (tq ? kj : 0);
          const double bu = __shfl(Q.tq_up, src, 16), bl = __shfl(Q.tq_lo, src, 16);
          b_p = tq ? -(lower ? bl : bu) : b_p;
        }
        const double np_tab = sign * lds_nrm[64 * tab_kind + ((int)threadIdx.x & 63)];
        npj = (has && pleg == leg) ? np_tab : 0.0;
        // directions as in a pass; z'n_p of a row without a candidate is 0: biased to 1, its step is 0
        double za[3] = {0.0, 0.0, 0.0}, ra[3] = {0.0, 0.0, 0.0};
        static_for<kV>([&](auto J) {
          constexpr int j = J;
          fmac_bc<lane_of(j), j == 0>(za[j % 3], npj, H[j]);
          fmac_bc<lane_of(j)>(ra[j % 3], npj, Ns[j]);
        });
        const double zw = (za[0] + za[1]) + za[2], rw = (ra[0] + ra[1]) + ra[2];
        const double znw = row_sum(zw * npj);
        const double s_p = row_sum(npj * x) - b_p; // slack of the row at x
        // a row that depends on the rows installed before it is left out: z'n_p is then rounding noise, which with the entries
        // of H reaching 1 / w_reg = 1e4 means up to 1e-10, while an independent row has z'n_p >= |n|^2 / trace(G) ~ 1e-3
        const bool ok = has && znw > 1e-6;
        const double zi = rcp_nr(sel(ok, znw, 1.0));
        const double tw = sel(ok, -s_p * zi, 0.0);
        x = fma(tw, zw, x);
        u = fma(-tw, rw, u);
        const int newlane = __ffs(~used & 0xFFFu) - 1;
        const bool newslot = ok && lr == newlane;
        vec = sel(ok, zw * zi, 0.0);
        hc = sel(ok, -zw, 0.0);
        nc = sel(newslot, 1.0, sel(ok, -rw, 0.0));
        u = sel(newslot, tw, u);
        idk = newslot ? p : idk;
        used |= ok ? (1u << newlane) : 0u;
        act_mask |= ok ? (one << p) : 0;
        rnorm2 = sel(ok, vmax(rnorm2, znw), rnorm2);
        q += ok ? 1 : 0;
        warm_updates += ok ? 1 : 0;
        update_only();
      }
    }
    if (by_rounds) warm_updates = q; // (installs only so far: one counter in the rounds)
    for (;;) { // at most q rounds: every round frees a slot and none is taken
      const bool slot = (used & lanebit) != 0u;
      const double umin = row_min(sel(slot, u, inf));
      const bool neg = umin < 0.0;
      if (__builtin_amdgcn_ballot_w64(neg) == 0ull) break;
      const int lpos = neg ? row_first(slot && u == umin) : 0;
      const double uk = __shfl(u, lpos, 16);
      const int drop_id = drop_vectors(lpos); // vec = n~, hc = n~ / e, nc = -(N* G n~) / e (-1 on the slot's own lane)
      // x -= (n~ / e) u_k, u -= (N* G n~ / e) u_k: the minimiser and the multipliers without the dropped row
      x = sel(neg, fma(-hc, uk, x), x);
      u = sel(neg, fma(nc, uk, u), u);
      vec = sel(neg, vec, 0.0); hc = sel(neg, hc, 0.0); nc = sel(neg, nc, 0.0);
      act_mask &= neg ? ~(one << drop_id) : ~(mask_t)0;
      used &= neg ? ~(1u << lpos) : ~0u;
      q -= neg ? 1 : 0;
      warm_updates += neg ? 1 : 0;
      update_only();
    }
    npj = 0.0;
    // (parked in the robot's LDS block until the loop is over: one register less across it -- the 168-register form spills
    // otherwise)
    if (lr == 0) reinterpret_cast<int *>(lds_row + kWarmSlot)[0] = warm_updates;
    warm_updates = 0;
  }
  QL_QP_BLOCK_STAMP(2);
  {
    // lanes that are not here (rows that have left with kStatusNotPd) count as finished: ballots never see them
    done_m = ~__builtin_amdgcn_ballot_w64(true);
    const bool fin0 = update_and_select(std::integral_constant<int, 0>{}, true, true);
    note_finished(fin0);
  }

  // Loop structure.  A pass = directions and step lengths, then what the step is.  Passes in which EVERY live row adds
  // its constraint run in an inner loop that is one straight path: unpredicated bookkeeping, then update + selection.
  // It is left as soon as some row drops, fails or is infeasible; that pass is finished by the tail below (rows that
  // drop: unpredicated, no selection; rows that add: the same straight path as in the inner loop; a row that does
  // neither sends the whole pass through the general predicated form) and the inner loop is entered again.
  // What the register allocator needs of this shape (found the hard way, tools/kernel_isa.py shows it at once): nothing
  // a pass computes may be live across a back edge, and the flags the branches test must be scalar masks assigned where
  // control is uniform -- a lane-mask boolean that lives across the loops, a second site that updates the finished mask
  // inside one branch of the tail, or an exit out of both loops at once each cost 24 register copies of H and N* per pass
  // or turn every branch into execution-mask bookkeeping.
  // Terminates: at most kMaxOuter adds, every drop undoes an earlier add, a failed add bans its row until the next add.
  double z = 0.0, r = 0.0, zn = 0.0, zinv = 0.0, t = 0.0, tl1 = 0.0, tl2 = 0.0, ratio = 0.0;
  // ---- a full step that adds the candidate: H -= z z'/d, N* <- [N* - r z'/d ; z'/d], the new row goes to the lowest
  // free slot lane; then the selection of the next candidate in the shadow of the update
  const auto add_step = [&]() -> bool {
    x += t * z;
    u = fma(-t, r, u);
    const int newlane = __ffs(~used & 0xFFFu) - 1;
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
  // ---- a partial step (t1 < t2), or a dual step only when t2 is infinite: the blocking constraint leaves the working set
  // and the same candidate continues
  const auto drop_step = [&]() {
    const double tp = (tl2 >= inf) ? 0.0 : t;
    x += tp * z;
    u = fma(-t, r, u);
    ucand += t;
    sp += tp * zn; // slack of ip after a partial step (:436-440, linear in t)
    const int lpos = row_first(ratio == tl1 && ratio < inf);
    const double r_lpos = __shfl(r, lpos, 16);
    const int drop_id = drop_vectors(lpos);
    act_mask &= ~(one << drop_id);
    used &= ~(1u << lpos);
    q--;
    update_only();
    // the same candidate continues with the working set one smaller: with H' = H + n~ n~'/e and N*' = N* - c n~'/e
    // its directions are z' = z + (n~/e) r_k, r' = r - (c/e) r_k, z'n = zn + r_k^2/e (r_k = n~'n_p = r of the
    // dropped slot) -- no need for the 24 broadcasts of the next pass when every live row has dropped
    z = fma(hc, r_lpos, z);
    r = fma(nc, r_lpos, r);
    zn = fma(r_lpos * r_lpos, drop_einv, zn);
  };
  // ---- directions of a pass: z = H n_p (lane i), r = N* n_p (slot lane k; 0 on free lanes, whose rows are 0), z'n_p
  // three partial sums per product (z and r alternate: consecutive dependent FMAs are 6 instructions apart)
  // (one or two accumulators per product instead of three -- four moves and adds fewer, the compiler fills the dependent
  // pairs with s_nop -- measured the same to 0.1 us on one box: profiles/r5/ab_dirs_accumulators.txt)
  const auto general_dirs = [&]() {
    double za[3] = {0.0, 0.0, 0.0}, ra[3] = {0.0, 0.0, 0.0};
    static_for<kV>([&](auto J) {
      constexpr int j = J;
      fmac_bc<lane_of(j), j == 0>(za[j % 3], npj, H[j]);
      fmac_bc<lane_of(j)>(ra[j % 3], npj, Ns[j]);
    });
    z = (za[0] + za[1]) + za[2];
    r = (ra[0] + ra[1]) + ra[2];
    zn = row_sum(fma(z, npj, zb));
  };
  // ---- step lengths, QuadProg++.cc:304-331; returns whether the pass is a full step that adds the candidate
  const auto step_lengths = [&]() -> bool {
    const bool slot = (used & lanebit) != 0u;
    const float zf = (float)z;
    const double zz = (double)row_sum_f32(zf * zf); // only compared with eps below
    const double ur = u * rcp_nr1(r);
    ratio = sel(slot && r > 0.0, ur, inf);
    tl1 = row_min(ratio);
    zinv = rcp_nr(zn);
    const double t2v = -sp * zinv;
    const bool exhausted = q >= 3 * nS; // empty null space: z is exactly 0 in the reference
    tl2 = sel((int)(!exhausted) & (int)(fabs(zz) > eps) & (int)(!(t2v < 0.0)), t2v, inf);
    t = vmin(tl1, tl2);
    // a full step (:384) whose constraint can be added (:392: |R_qq| = sqrt(z'n_p) > eps * R_norm, compared squared)
    return (tl2 < inf) && (tl2 <= tl1) && (zn > eps * eps * rnorm2);
  };
  // ---- the pass of a live row in general form (all row-uniform): a degenerate add, a dual step only, an infeasible problem
  const auto general_pass = [&](bool is_add) -> bool {
    // ---- the pass of the live rows in general form (all row-uniform)
    const bool infeasible = !(t < inf);                          // :339-344
    const bool dual_only = (tl2 >= inf);
    const bool full = !infeasible && !dual_only && (tl2 <= tl1);   // :384
    const bool degenerate = full && !is_add;
    const bool is_drop = !infeasible && !full;                   // partial or dual-only step
    if (infeasible) status = kStatusInfeasible;
    const double tp = (infeasible || dual_only || degenerate) ? 0.0 : t;
    const double td = (infeasible || degenerate) ? 0.0 : t;
    x += tp * z;
    u = fma(-td, r, u);
    ucand += td;
    sp += tp * zn;
    // add (predicated).  A numerically dependent normal is skipped and selection repeated.
    const int newlane = __ffs(~used & 0xFFFu) - 1;
    const bool newslot = is_add && (lr == newlane);
    vec = is_add ? z * zinv : 0.0;
    hc = is_add ? -z : 0.0;
    nc = sel(newslot, 1.0, sel(is_add, -r, 0.0));
    u = newslot ? ucand : u;
    idk = newslot ? ip : idk;
    used |= is_add ? (1u << newlane) : 0u;
    act_mask |= is_add ? (one << ip) : 0;
    rnorm2 = is_add ? vmax(rnorm2, zn) : rnorm2;
    q += is_add ? 1 : 0;
    excl |= degenerate ? (one << ip) : 0;
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
  // Which rows have finished (done_m), add (add_m) or drop (drop_m) in the pass at hand are scalar 64-bit masks and the
  // branches compare them; the finished rows ride along on every path (ghost rows, above).
  const auto in = [](unsigned long long m) -> bool { return __builtin_amdgcn_inverse_ballot_w64(m); };
  if (~done_m != 0ull) {
    for (;;) {
      unsigned long long add_m = 0ull;
      for (;;) { // passes in which every live row adds
        general_dirs();
        add_m = __builtin_amdgcn_ballot_w64(step_lengths());
        if (~(add_m | done_m) != 0ull) break;
        note_finished(add_step());
        if (~done_m == 0ull) break;
      }
      if (~done_m == 0ull) break;
      // ---- a pass in which some live row cannot add yet.  Every pass ends with an add for every live row: the rows whose
      // step is blocked drop the blocking constraint first -- the straight drop path, then the step lengths again from the
      // continued directions, repeated while any of them is still blocked -- and then all live rows add together.  (Until
      // round 3 such a pass did one step per row: the rows that dropped came back for their add in the next pass, and a
      // wavefront paid 1.03 us for it where this form pays 0.5 us per round of drops: the slowest wavefront of the
      // survey-literal batch 19.8 -> 16.9 us above the floor, tools/experiments/lockstep_schemes.py.)  A row that does
      // neither (a degenerate add, a dual step only, an infeasible problem) takes the general predicated form and sits
      // the rest of the pass out.
      unsigned long long gen_m = 0ull; // rows that took the general form in this pass
      bool fin = false;
      for (;;) {
        const unsigned long long live_m = ~done_m & ~gen_m;
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
        fin = fin || (fin_add && in(adders_m));
      }
      note_finished(fin);
      if (~done_m == 0ull) break;
    }
  }
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    x = lds_row[lr];
    const int2 a = reinterpret_cast<const int2 *>(lds_row + 16)[lr], b = reinterpret_cast<const int2 *>(lds_row + 32)[lr];
    used = (unsigned)a.x; idk = a.y; q = b.x & 255; status = b.y;
    if constexpr (kWarm) warm_updates = reinterpret_cast<const int *>(lds_row + kWarmSlot)[0];
    iters_out = (b.x >> 8) + warm_updates; // what this robot cost: the installs and drops of a warm start count as passes
  }
  if constexpr (kWarm) { // the final working set as a bit mask: the OR over the slot lanes of a row
    const mask_t mine = ((used >> lr) & 1u) ? (one_v<mask_t> << idk) : 0;
    unsigned lo = (unsigned)mine, hi = 0u;
    if constexpr (kTorque) hi = (unsigned)((unsigned long long)mine >> 32);
    static_for<4>([&](auto K) {
      constexpr int ctrl = K == 0 ? 0x128 : K == 1 ? 0x124 : K == 2 ? 0x122 : 0x121;
      lo |= (unsigned)__builtin_amdgcn_mov_dpp((int)lo, ctrl, 0xF, 0xF, true);
      if constexpr (kTorque) hi |= (unsigned)__builtin_amdgcn_mov_dpp((int)hi, ctrl, 0xF, 0xF, true);
    });
    if (ws_out) *ws_out = status == kStatusOk ? ((unsigned long long)hi << 32 | lo) : 0ull;
  }

  QL_STAMP(7); QL_QP_BLOCK_STAMP(7);
  // ---------------------------------------------------------------- refinement on the final working set
  // (a warm start that installed rows and dropped them all again ends with an empty set and operators that have drifted all
  // the same: it is refined like any other)
  if (status == kStatusOk && (q > 0 || (kWarm && warm_updates > 0))) {
    // export N* through LDS once: lane (leg,c) needs column myidx of N*
    if (lr < kV) {
#pragma unroll
      for (int j = 0; j < kV; j++) lds_row[12 * lr + j] = Ns[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double NsT[kV]; // N*[k][myidx], k = 0 .. kV-1
#pragma unroll
    for (int k = 0; k < kV; k++) NsT[k] = comp ? lds_row[12 * k + myidx] : 0.0;
    // the row behind my slot: leg, kind, and the lane that watches it
    const int lg = kTorque ? ((idk * 47) >> 9) : id_leg(idk), tt = idk - kKinds * lg;
    const bool myslot = (used >> lr) & 1u;
    const int src = myslot ? (4 * lg + (tt == 0 ? 0 : (tt < 5 ? tt - 1 : (tt - 5) >> 1))) : 0;
    // The explicit operators drift with every rank-one update; one pass is enough for the ~30 updates of the longest cold
    // start (6.3 M control steps against the oracle: 4e-8).  A warm start from a set that does not fit -- installs and drops
    // of rows that a cold start would never have touched -- can do more updates than that: a pass more per 16 of them.
    int passes = Q.refine_passes;
    if constexpr (kWarm) passes += warm_updates > 0 ? iters_out >> 4 : 0; // (no set handed in: the cold start, bit for bit)
    for (int pass = 0; pass < passes; pass++) {
      // (1) reduced gradient: x -= H (G x + g0)
      double grad = g0;
      static_for<kV>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(grad, x, Gm[j]); });
      double corr = 0.0;
      static_for<kV>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(corr, grad, H[j]); });
      x -= corr;
      // (2) constraint residuals rho_k = b_k - n_k'x on slot lanes; x += N*' rho
      double s_min, s_fric;
      slacks(x, s_min, s_fric);
      const double vm = __shfl(s_min, src, 16), vf = __shfl(s_fric, src, 16);
      double res = sel(tt == 0, vm, vf);
      if constexpr (kTorque) {
        const double vu = __shfl(s_up, src, 16), vl = __shfl(s_lo, src, 16);
        res = sel(tt < 5, res, sel(((tt - 5) & 1) != 0, vl, vu));
      }
      const double rho = sel(myslot, -res, 0.0);
      double dx = 0.0;
      static_for<kV>([&](auto K) { constexpr int k = K; fmac_bc<k, k == 0>(dx, rho, NsT[k]); });
      x += dx;
      if constexpr (kWarm) {
        // A pass shrinks the error by the operators' relative drift (<= 1e-3: one pass takes the first correction of a cold
        // start, <= 4e-5 over 6.3 M steps, to 4e-8).  A set that fitted leaves a first correction of up to a few 1e-4 when it
        // was installed by rounds: one pass, as for a cold start.  A set that did not can leave x far from the optimum of its
        // final set; such a robot passes again until the correction is below kRefineAgain = 1e-3 (what is left then is
        // <= 1e-6 of it; with 1e-4, round 5's value, one static robot in three of the bench batch passed twice: 14.6 -> 13.9 us
        // per step, the soak on trajectories with jumps -- 4.7 M steps -- the same 6e-8: profiles/r6/ab_refine_again.txt).
        const double moved = -row_min(-(__builtin_fabs(corr) + __builtin_fabs(dx)));
        passes += (warm_updates > 0 && pass + 1 == passes && passes < kMaxRefinePasses && moved > kRefineAgain) ? 1 : 0;
      }
    }
  }

  if constexpr (kWarm) {
    // A warm start is taken on trust only as far as its answer checks out: at the final x every row must hold and every
    // multiplier of the final set, u = N* (G x + g0), must be non-negative (to 1e-6 N: the efforts' own tolerance).  The
    // method guarantees both in exact arithmetic; a set that has nothing to do with this robot's state (a reset robot, a
    // caller's bug) can cost so many installs and drops that the explicit operators drift out of it.  Such a robot is
    // reported (QLAMD_STATUS_WARM_REJECTED: outputs as for any failed solve, working set 0, so that its next step starts
    // cold) rather than solved a second time here: a retry inside the kernel keeps the whole problem live across the loop,
    // which the 168-register form pays for with spills (measured: 88-556 bytes of scratch in three arrangements).
    if (__builtin_amdgcn_ballot_w64(status == kStatusOk && warm_updates > 0) != 0ull) {
      double s_min, s_fric;
      slacks(x, s_min, s_fric);
      double worst = vmin(s_fric, sel(c == 0, s_min, s_fric));
      if constexpr (kTorque) worst = sel(comp, vmin(worst, vmin(s_up, s_lo)), worst);
      const double wmin = row_min(sel(on, worst, 0.0));
      double grad = g0;
      static_for<kV>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(grad, x, Gm[j]); });
      double ua[3] = {0.0, 0.0, 0.0};
      static_for<kV>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(ua[j % 3], grad, Ns[j]); });
      const bool myslot2 = ((used >> lr) & 1u) != 0u;
      const double umin = row_min(sel(myslot2, (ua[0] + ua[1]) + ua[2], 0.0));
      const bool accept = wmin > -1e-6 && umin > -1e-6;
      status = (status == kStatusOk && warm_updates > 0 && !accept) ? kStatusWarmRejected : status;
    }
  }
  return status;
}

} // namespace coop
} // namespace qlamd
