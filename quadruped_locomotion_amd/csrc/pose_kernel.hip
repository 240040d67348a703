// HIP kernels (gfx950) of the pose optimisation (rows a16/a17, f3) and of the dense QP batch (rows a14/a15) and their
// part of the C-ABI of include/qlamd.h.
#include "pose_coop.hpp"
#include "qp_coop.hpp"
#include "pose_core.hpp"
#include "context.hpp"

using namespace qlamd;
using namespace qlamd::rt;

namespace {

// ---- pose optimisation: problem records --------------------------------------------------
struct PosePtrs {
  const double *stance, *nominal, *polygon, *rcom, *maxlen, *pose;
  const uint8_t *mask;
  const int32_t *nverts;
};

// Every load of a problem is issued unconditionally and before the first use: an absent optional array is read
// through `stance` (always present and at least as long) and its value replaced afterwards, so that no load sits
// behind a branch and the whole problem costs one memory round trip.
__device__ __forceinline__ void load_pose_problem(const PoseParamsDev &P, const PosePtrs &s, int64_t i, PoseProblem &pb,
                                                  double pose[7]) {
  const uint8_t *maskp = s.mask ? s.mask : reinterpret_cast<const uint8_t *>(s.stance);
  const double *rcomp = s.rcom ? s.rcom : s.stance;
  const int32_t *nvp = s.nverts ? s.nverts : reinterpret_cast<const int32_t *>(s.stance);
  const double *posep = s.pose ? s.pose : s.stance;
  const uint32_t m4 = *reinterpret_cast<const uint32_t *>(maskp + 4 * i);
  const int32_t nv = nvp[i];
  double rc[3], ps[7], poly[8];
#pragma unroll
  for (int a = 0; a < 3; a++) rc[a] = rcomp[3 * i + a];
#pragma unroll
  for (int a = 0; a < 7; a++) ps[a] = posep[7 * i + a];
  {
    const double2 *p2 = reinterpret_cast<const double2 *>(s.polygon + 8 * i);
#pragma unroll
    for (int k = 0; k < 4; k++) { const double2 v = p2[k]; poly[2 * k] = v.x; poly[2 * k + 1] = v.y; }
  }
  pose_problem_load_legs(
      P, pb, [&](int l, int a) { return s.stance[12 * i + 3 * l + a]; },
      [&](int l, int a) { return s.nominal[12 * i + 3 * l + a]; }, [&](int l) { return s.maxlen[4 * i + l]; },
      [&]() {
        unsigned limb_mask = 0xFu;
        if (s.mask) {
          limb_mask = 0;
#pragma unroll
          for (int l = 0; l < 4; l++) limb_mask |= ((m4 >> (8 * l)) & 0xFFu) ? (1u << l) : 0u;
        }
        return limb_mask;
      });
#pragma unroll
  for (int l = 0; l < 4; l++) { pb.polygon[l][0] = poly[2 * l]; pb.polygon[l][1] = poly[2 * l + 1]; }
#pragma unroll
  for (int a = 0; a < 3; a++) pb.r_com[a] = s.rcom ? rc[a] : 0.0;
  pb.n_vertices = s.nverts ? nv : 4;
#pragma unroll
  for (int a = 0; a < 7; a++) pose[a] = s.pose ? ps[a] : (a == 3 ? 1.0 : 0.0);
}

// The same record fetched by the 16 lanes of a row together: every lane loads at most one element of each array (all loads
// in flight before the first is consumed: one memory round trip for the whole wavefront), stores it at its place of the
// record in LDS -- the per-leg arrays permuted into iteration order as above -- and the row reads the record back after a
// wavefront-level fence (a row reads only what its own lanes wrote: no barrier).  One lane doing it alone, as
// load_pose_problem does for the one-problem-per-lane kernels, took 2.1 us of a 14.5 us launch (tools/stamp_probe_pose.py).
__device__ __forceinline__ void load_pose_problem_row(const PoseParamsDev &P, const PosePtrs &s, int64_t i, PoseProblem &pb,
                                                      double *pose_lds) {
  const int lr = threadIdx.x & 15;
  const uint8_t *maskp = s.mask ? s.mask : reinterpret_cast<const uint8_t *>(s.stance);
  const double *rcomp = s.rcom ? s.rcom : s.stance;
  const int32_t *nvp = s.nverts ? s.nverts : reinterpret_cast<const int32_t *>(s.stance);
  const double *posep = s.pose ? s.pose : s.stance;
  // lanes 0..11: slot k = lr / 3, component a; the limb behind the slot.  (The batch constants are pinned in scalar
  // registers first: left as loads from the parameter block, the selects below are folded into ONE indexed load, which
  // sends the whole block to scratch memory.)
  int lo[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) { lo[kk] = P.leg_order[kk]; asm volatile("" : "+s"(lo[kk])); }
  const int k = lr < 12 ? (lr * 11) >> 5 : 3, a = lr < 12 ? lr - 3 * k : 0;
  const int l = k == 0 ? lo[0] : k == 1 ? lo[1] : k == 2 ? lo[2] : lo[3];
  const int k4 = lr & 3;
  const int l4 = k4 == 0 ? lo[0] : k4 == 1 ? lo[1] : k4 == 2 ? lo[2] : lo[3];
  const double st = s.stance[12 * i + 3 * l + a], nm = s.nominal[12 * i + 3 * l + a];
  const double pl = s.polygon[8 * i + (lr & 7)];
  const double ml = s.maxlen[4 * i + l4];
  const double rc = rcomp[3 * i + (lr < 3 ? lr : 2)], ps = posep[7 * i + (lr < 7 ? lr : 6)];
  const uint32_t m4 = *reinterpret_cast<const uint32_t *>(maskp + 4 * i);
  const int32_t nv = nvp[i];
  // hip of my slot's limb, component a (batch constants: selected, not loaded)
  double hp = 0.0;
#pragma unroll
  for (int ll = 0; ll < 4; ll++) {
    double h0 = P.hips[ll][0], h1 = P.hips[ll][1], h2 = P.hips[ll][2];
    asm volatile("" : "+s"(h0), "+s"(h1), "+s"(h2));
    const double h = a == 0 ? h0 : a == 1 ? h1 : h2;
    hp = l == ll ? h : hp;
  }
  unsigned limb_mask = 0xFu, present = 0u;
  if (s.mask) {
    limb_mask = 0u;
#pragma unroll
    for (int ll = 0; ll < 4; ll++) limb_mask |= ((m4 >> (8 * ll)) & 0xFFu) ? (1u << ll) : 0u;
  }
#pragma unroll
  for (int kk = 0; kk < 4; kk++) present |= ((limb_mask >> lo[kk]) & 1u) ? (1u << kk) : 0u;
  if (lr < 12) {
    (&pb.stance[0][0])[lr] = st;
    (&pb.nominal[0][0])[lr] = nm;
    (&pb.hips[0][0])[lr] = hp;
  }
  if (lr < 8) (&pb.polygon[0][0])[lr] = pl;
  if (lr < 4) pb.max_len[lr] = ml;
  if (lr < 3) pb.r_com[lr] = s.rcom ? rc : 0.0;
  if (lr < 7) pose_lds[lr] = s.pose ? ps : (lr == 3 ? 1.0 : 0.0);
  if (lr == 0) {
    pb.n_vertices = s.nverts ? nv : 4;
    pb.present = present;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);
}

// Lane-cooperative form (csrc/pose_coop.hpp): 16 lanes per problem, 4 problems per wavefront.
__global__ __launch_bounds__(64) void pose_sqp_coop_kernel(const PoseParamsDev P, const PosePtrs s, int64_t B,
                                                           double *__restrict__ pose_out, int32_t *__restrict__ iters,
                                                           int32_t *__restrict__ status) {
  __shared__ PoseProblem pbs[coop::kPoseCoopRows];
  __shared__ double pose0[coop::kPoseCoopRows][8];
  __shared__ double rows[coop::kPoseCoopRows * coop::kPoseCoopLdsDoubles];
  const int row = threadIdx.x >> 4, lr = threadIdx.x & 15;
  int64_t i = (int64_t)blockIdx.x * coop::kPoseCoopRows + row;
  const bool live = i < B;
  if (!live) i = B - 1;
  QL_STAMP(16);
  load_pose_problem_row(P, s, i, pbs[row], pose0[row]);
  double pose[7];
#pragma unroll
  for (int a = 0; a < 7; a++) pose[a] = pose0[row][a];
  int it = 0;
  QL_STAMP(17);
  const int st = coop::pose_sqp_coop(P, pbs[row], live, rows + row * coop::kPoseCoopLdsDoubles, pose, it);
  QL_STAMP(29);
  if (lr == 0 && live) {
#pragma unroll
    for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
    if (iters) iters[i] = it;
    status[i] = st;
  }
  QL_STAMP(30);
}

// PoseOptimizationQP (position only) in the row layout, and PoseConstraintsChecker (one lane per problem)
__global__ __launch_bounds__(64) void pose_qp_coop_kernel(const PoseParamsDev P, const PosePtrs s, int64_t B,
                                                          double *__restrict__ pose_out, int32_t *__restrict__ status) {
  __shared__ PoseProblem pbs[coop::kPoseCoopRows];
  __shared__ double pose0[coop::kPoseCoopRows][8];
  __shared__ double rows[coop::kPoseCoopRows * coop::kPoseCoopLdsDoubles];
  const int row = threadIdx.x >> 4, lr = threadIdx.x & 15;
  int64_t i = (int64_t)blockIdx.x * coop::kPoseCoopRows + row;
  const bool live = i < B;
  if (!live) i = B - 1;
  load_pose_problem_row(P, s, i, pbs[row], pose0[row]);
  double pose[7];
#pragma unroll
  for (int a = 0; a < 7; a++) pose[a] = pose0[row][a];
  const int st = coop::pose_qp_coop(P, pbs[row], live, rows + row * coop::kPoseCoopLdsDoubles, pose);
  if (lr == 0 && live) {
#pragma unroll
    for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
    status[i] = st;
  }
}

__global__ __launch_bounds__(64) void pose_check_kernel(const PoseParamsDev P, const PosePtrs s,
                                                        const double *__restrict__ min_len, double leg_tol, int64_t B,
                                                        uint8_t *__restrict__ ok) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= B) return;
  PoseProblem pb;
  double pose[7], mn[4];
  load_pose_problem(P, s, i, pb, pose);
#pragma unroll
  for (int k = 0; k < 4; k++) mn[k] = min_len ? min_len[4 * i + P.leg_order[k]] : 0.0;
  ok[i] = pose_check(pb, pose, mn, leg_tol) ? 1 : 0;
}

__device__ __forceinline__ void load_sfo(const PosePtrs &s, const double *__restrict__ sfo_in, int64_t i, double sfo[4][3]) {
  const double *src = sfo_in ? sfo_in : s.stance; // default: the stance itself (all four limbs needed, :76-77)
#pragma unroll
  for (int l = 0; l < 4; l++)
#pragma unroll
    for (int a = 0; a < 3; a++) sfo[l][a] = src[12 * i + 3 * l + a];
}

// PoseOptimizationGeometric on its own: one problem per lane (pose_core.hpp's serial form: Kabsch matrix, cyclic Jacobi,
// heading and roll / pitch).  No QP is involved and nothing is shared between the lanes of a row, so the row layout has
// nothing to offer here: the 16-lane form of the same arithmetic (coop::pose_geometric_coop, the first stage of
// base_auto_coop_kernel) measured 37 us as its own launch against 14 us for this one at 4096 problems.
__global__ __launch_bounds__(64) void pose_geometric_kernel(const PoseParamsDev P, const PosePtrs s,
                                                            const double *__restrict__ sfo_in, int64_t B,
                                                            double *__restrict__ pose_out) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= B) return;
  PoseProblem pb;
  double pose[7], sfo[4][3];
  load_pose_problem(P, s, i, pb, pose);
  load_sfo(s, sfo_in, i, sfo);
  pose_geometric(pb, sfo, pose);
#pragma unroll
  for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
}

// BaseAuto::optimizePose: geometric -> QP -> check -> SQP for the problems the check rejects
__global__ __launch_bounds__(64) void base_auto_coop_kernel(const PoseParamsDev P, const PosePtrs s,
                                                            const double *__restrict__ sfo_in,
                                                            const double *__restrict__ min_len, double leg_tol, int64_t B,
                                                            double *__restrict__ pose_out, int32_t *__restrict__ stage,
                                                            int32_t *__restrict__ iters, int32_t *__restrict__ status) {
  __shared__ PoseProblem pbs[coop::kPoseCoopRows];
  __shared__ double io[coop::kPoseCoopRows][24];
  __shared__ double rows[coop::kPoseCoopRows * coop::kPoseCoopLdsDoubles];
  const int row = threadIdx.x >> 4, lr = threadIdx.x & 15;
  int64_t i = (int64_t)blockIdx.x * coop::kPoseCoopRows + row;
  const bool live = i < B;
  if (!live) i = B - 1;
  {
    // the record by the 16 lanes of the row (load_pose_problem_row), the stance for orientation (lanes 0..11, limb order)
    // and the minimal lengths (lanes 0..3, iteration order) beside it: io = pose (7) | sfo (12) | min_len (4)
    const double *src = sfo_in ? sfo_in : s.stance; // default: the stance itself (all four limbs needed, :76-77)
    const double sf = src[12 * i + (lr < 12 ? lr : 0)];
    int lo[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) { lo[kk] = P.leg_order[kk]; asm volatile("" : "+s"(lo[kk])); }
    const int k4 = lr & 3;
    const int l4 = k4 == 0 ? lo[0] : k4 == 1 ? lo[1] : k4 == 2 ? lo[2] : lo[3];
    const double mnl = (min_len ? min_len : s.maxlen)[4 * i + l4];
    load_pose_problem_row(P, s, i, pbs[row], io[row]);
    if (lr < 12) io[row][7 + lr] = sf;
    if (lr < 4) io[row][19 + lr] = min_len ? mnl : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  double pose[7], sfo[4][3], mn[4];
#pragma unroll
  for (int l = 0; l < 4; l++)
#pragma unroll
    for (int a = 0; a < 3; a++) sfo[l][a] = io[row][7 + 3 * l + a];
#pragma unroll
  for (int k = 0; k < 4; k++) mn[k] = io[row][19 + k];
  int stg = 0, it = 0;
  const int st = coop::base_auto_coop(P, pbs[row], sfo, mn, leg_tol, live, rows + row * coop::kPoseCoopLdsDoubles, pose, stg, it);
  if (lr == 0 && live) {
#pragma unroll
    for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
    if (stage) stage[i] = stg;
    if (iters) iters[i] = it;
    status[i] = st;
  }
}

// Lane-cooperative dense QP batch (csrc/qp_coop.hpp): 16 lanes per problem, 4 problems per wavefront.
// N = 6 for n <= 6, N = 12 otherwise; KC = 2 inequalities per lane for m <= 24, 3 for m <= 48; up to two equality
// columns.
template <int N, int KC>
__global__ __launch_bounds__(64) void qp_coop_kernel(int n, int p, int m, const double *__restrict__ G,
                                                     const double *__restrict__ g0, const double *__restrict__ CE,
                                                     const double *__restrict__ ce0, const double *__restrict__ CI,
                                                     const double *__restrict__ ci0, int64_t B, double *__restrict__ x,
                                                     double *__restrict__ obj, int32_t *__restrict__ status, const PlacePtrs pp) {
  typedef coop::QpCoopLds<N, KC> L;
  __shared__ double rows[coop::kQpCoopRows * L::kTotal];
  const int row = threadIdx.x >> 4, lr = threadIdx.x & 15;
  bool live;
  const int64_t i = placed_index(pp, (int64_t)blockIdx.x * coop::kQpCoopRows + row, B, live);
  // every load is issued unconditionally with a clamped index; values outside the problem are replaced afterwards
  const int rv = lr < n ? lr : 0;               // my variable (row of G)
  int cs[KC];                                   // my inequalities
  bool v[KC];
#pragma unroll
  for (int s = 0; s < KC; s++) { v[s] = lr + 16 * s < m; cs[s] = v[s] ? lr + 16 * s : 0; }
  double Gm[N], a[KC][N], b[KC];
  const double *Gp = G + (size_t)i * n * n + (size_t)rv * n;
  const double *Cp = CI ? CI + (size_t)i * n * m : G;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int kk = k < n ? k : 0;
    Gm[k] = Gp[kk];
#pragma unroll
    for (int s = 0; s < KC; s++) a[s][k] = Cp[(size_t)kk * (m > 0 ? m : 1) + cs[s]];
  }
  double gl = g0[(size_t)i * n + rv];
#pragma unroll
  for (int s = 0; s < KC; s++) b[s] = m > 0 ? ci0[(size_t)i * m + cs[s]] : 0.0;
  double ne = p > 0 ? CE[((size_t)i * n + rv) * p] : 0.0, e0 = p > 0 ? ce0[(size_t)i * p] : 0.0;
  double ne2 = p > 1 ? CE[((size_t)i * n + rv) * p + 1] : 0.0, e02 = p > 1 ? ce0[(size_t)i * p + 1] : 0.0;
  const bool var = lr < n;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const bool in = var && k < n;
    Gm[k] = in ? Gm[k] : ((lr == k && lr < N) ? 1.0 : 0.0); // identity padding for rows n..N-1
#pragma unroll
    for (int s = 0; s < KC; s++) a[s][k] = (v[s] && k < n) ? a[s][k] : 0.0;
  }
  gl = var ? gl : 0.0;
  ne = var ? ne : 0.0;
  ne2 = var ? ne2 : 0.0;
#pragma unroll
  for (int s = 0; s < KC; s++) b[s] = v[s] ? b[s] : 0.0;
  double xo, fo;
  int its = 0;
  const int st = coop::qp_coop_impl<N, KC>(Gm, gl, n, n, m, p > 0, ne, e0, a, b, v, !live, rows + row * L::kTotal, xo, fo,
                                           p > 1, ne2, e02, nullptr, nullptr, 0, -1, nullptr, &its);
  if (live) {
    if (var) x[(size_t)i * n + lr] = xo;
    if (lr == 0) {
      if (obj) obj[i] = fo;
      status[i] = st;
      if (pp.iterations) pp.iterations[i] = its;
    }
  }
}

// Weighted least squares with equality rows and two-sided inequality rows, the form
// ooqpei::QuadraticProblemFormulation::solve(A, S, b, W, C, c, D, d, f, x) takes (call sites
// ContactForceDistribution.cpp:367,490):   min (Ax - b)'S(Ax - b) + x'Wx   s.t.  Cx = c,  d <= Dx <= f.
// Lane i < n assembles row i of A'SA + W and entry i of -A'Sb (the objective halved: same minimiser); row r of D
// gives the one-sided rows r (lower bound) and r + 24 (upper bound), a bound of +-DBL_MAX or +-inf being none;
// the equality rows are projected out one by one inside qp_coop_impl.
template <int N>
__global__ __launch_bounds__(64) void weighted_lsq_qp_kernel(int n, int k, int p, int m, const double *__restrict__ A,
                                                             const double *__restrict__ S, const double *__restrict__ bb,
                                                             const double *__restrict__ W, const double *__restrict__ C,
                                                             const double *__restrict__ cc, const double *__restrict__ D,
                                                             const double *__restrict__ dlo, const double *__restrict__ fup,
                                                             int64_t B, double *__restrict__ x, int32_t *__restrict__ status,
                                                             const PlacePtrs pp) {
  typedef coop::QpCoopLds<N, 3> L;
  __shared__ double rows[coop::kQpCoopRows * L::kTotal];
  __shared__ double eq_dirs[coop::kQpCoopRows * 12 * N]; // step directions of the equality rows (refinement sweep)
  const int row = threadIdx.x >> 4, lr = threadIdx.x & 15;
  bool live;
  const int64_t i = placed_index(pp, (int64_t)blockIdx.x * coop::kQpCoopRows + row, B, live);
  const bool var = lr < n;
  const int rv = var ? lr : 0;
  double Gm[N], gl = 0.0;
#pragma unroll
  for (int j = 0; j < N; j++) Gm[j] = 0.0;
  const double *Ap = A + (size_t)i * k * n;
  {
    // lane i holds column i of A (one load per row of A); entry (i, j) of A'SA is the sum over the rows r of
    // A[r][i] S[r] (mine) times A[r][j] (lane j's, through the DPP operand): k x N broadcast-FMAs, no further loads
    double acol[12], sa[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
      const bool in = r < k;
      const int rr = in ? r : 0;
      const double sr = S[(size_t)i * k + rr], br = bb[(size_t)i * k + rr];
      const double av = Ap[(size_t)rr * n + rv];
      acol[r] = (in && var) ? av : 0.0;
      sa[r] = acol[r] * sr;
      gl = fma(-sa[r], br, gl);
    }
    coop::static_for<12>([&](auto R) {
      constexpr int r = R;
      coop::static_for<N>([&](auto J) { constexpr int j = J; coop::fmac_bc<j, j == 0>(Gm[j], acol[r], sa[r]); });
    });
  }
  const double wi = W[(size_t)i * n + rv];
#pragma unroll
  for (int j = 0; j < N; j++) {
    const bool in = var && j < n;
    Gm[j] = in ? Gm[j] + (j == lr ? wi : 0.0) : ((lr == j && lr < N) ? 1.0 : 0.0); // identity padding for rows n..N-1
  }
  gl = var ? gl : 0.0;
  // one-sided rows of this lane: slots lr, lr + 16, lr + 32; slot j < 24 = lower bound of row j, else upper bound of row j - 24
  double a[3][N], b[3];
  bool v[3];
  const double big = 1.7976931348623157e308;
  int mine = 0;
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const int slot = lr + 16 * s, r = slot < 24 ? slot : slot - 24;
    const bool upper = slot >= 24, have = r < m;
    const int rr = have ? r : 0;
    const double bound = m > 0 ? (upper ? fup[(size_t)i * m + rr] : dlo[(size_t)i * m + rr]) : 0.0;
    v[s] = have && (upper ? bound < big : bound > -big);      // DBL_MAX / inf = no bound; NaN = no row
    b[s] = v[s] ? (upper ? bound : -bound) : 0.0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const double dj = (m > 0 && j < n) ? D[((size_t)i * m + rr) * n + j] : 0.0;
      a[s][j] = v[s] ? (upper ? -dj : dj) : 0.0;
    }
    mine += v[s] ? 1 : 0;
  }
  // rows that exist, for the termination tolerance (QuadProg++.cc:246 counts its m)
  int cnt = mine;
  cnt += __shfl_xor(cnt, 8, 16); cnt += __shfl_xor(cnt, 4, 16); cnt += __shfl_xor(cnt, 2, 16); cnt += __shfl_xor(cnt, 1, 16);
  double xo, fo;
  int st, its = 0;
  const double *Ci = C ? C + (size_t)i * p * n : nullptr, *ci = cc ? cc + (size_t)i * p : nullptr;
  // Slots 24..47 hold the upper bounds.  The reference's problem has none at all (f = DBL_MAX throughout,
  // ContactForceDistribution.cpp:246,329): when no problem of this wavefront has one, the two-rows-per-lane form of the
  // solver does with its 24 slots (the LDS block is laid out for the three-row form, the larger one).
  if (__builtin_amdgcn_ballot_w64(v[2] || (v[1] && lr >= 8)) == 0ull) {
    const double(&a2)[2][N] = reinterpret_cast<const double(&)[2][N]>(a);
    const double(&b2)[2] = reinterpret_cast<const double(&)[2]>(b);
    const bool(&v2)[2] = reinterpret_cast<const bool(&)[2]>(v);
    st = coop::qp_coop_impl<N, 2>(Gm, gl, n, n, 24, false, 0.0, 0.0, a2, b2, v2, !live, rows + row * L::kTotal, xo, fo, false, 0.0,
                                  0.0, Ci, ci, C ? p : 0, cnt, eq_dirs + row * 12 * N, &its);
  } else {
    st = coop::qp_coop_impl<N, 3>(Gm, gl, n, n, 48, false, 0.0, 0.0, a, b, v, !live, rows + row * L::kTotal, xo, fo, false, 0.0,
                                  0.0, Ci, ci, C ? p : 0, cnt, eq_dirs + row * 12 * N, &its);
  }
  if (live) {
    if (var) x[(size_t)i * n + lr] = xo;
    if (lr == 0) {
      status[i] = st;
      if (pp.iterations) pp.iterations[i] = its;
    }
  }
}

} // namespace

extern "C" {

void qlamd_pose_default_params(qlamd_pose_params *p) {
  if (!p) return;
  // free_gait_core/test/AdapterDummy.cpp:111-125 (same values as quadruped_state.cpp:83-97), LF RF RH LH
  const double hips[4][3] = {{0.42, 0.075, 0.0}, {0.42, -0.075, 0.0}, {-0.42, -0.075, 0.0}, {-0.42, 0.075, 0.0}};
  memcpy(p->hip_in_base, hips, sizeof(hips));
  p->com_weight = 2.0;     // PoseOptimizationObjectiveFunction.cpp:17
  p->tolerance = 0.05;     // PoseOptimizationSQP.cpp:99
  p->max_iterations = 30;
  p->dummy_equality = 1;   // sequencequadraticproblemsolver.cpp:25-26
  p->leg_order[0] = 2; p->leg_order[1] = 3; p->leg_order[2] = 1; p->leg_order[3] = 0;
}

// One driver for the pose entries; what differs per entry is the kernel and which optional arrays exist.
enum PoseMode { kPoseSqp = 0, kPoseQp = 1, kPoseCheck = 2, kPoseGeometric = 3, kPoseBaseAuto = 4 };
struct PoseCall {
  PoseMode mode;
  double *pose_out = nullptr;        // [B][7]   (all but check)
  int32_t *iterations = nullptr;     // [B]      (sqp, base_auto; optional)
  int32_t *status = nullptr;         // [B]      (sqp, qp, base_auto)
  int32_t *stage = nullptr;          // [B]      (base_auto; optional)
  uint8_t *ok = nullptr;             // [B]      (check)
  const double *min_len = nullptr;   // [B][4]   (check, base_auto; optional)
  const double *sfo = nullptr;       // [B][12]  (geometric, base_auto; optional -> stance)
  double leg_tol = 0.0;
};

static int pose_impl(const PoseCall &call, qlamd_context *ctx, const qlamd_pose_params *params,
                     const qlamd_pose_batch *in, int64_t batch, int memory, void *stream) {
  const PoseMode mode = call.mode;
  if (!ctx || !in || batch < 0) return QLAMD_ERR_INVALID_ARGUMENT;
  if (mode == kPoseCheck ? !call.ok : !call.pose_out) return QLAMD_ERR_INVALID_ARGUMENT;
  const bool has_status = mode == kPoseSqp || mode == kPoseQp || mode == kPoseBaseAuto;
  if (has_status && !call.status) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!params) return QLAMD_ERR_NOT_LOADED;
  if (!in->stance || !in->nominal_stance || !in->support_polygon || !in->max_limb_length) return QLAMD_ERR_INVALID_ARGUMENT;
  const bool needs_pose = mode != kPoseGeometric && mode != kPoseBaseAuto; // those two start from scratch
  if (needs_pose && !in->pose) return QLAMD_ERR_INVALID_ARGUMENT;
  if (params->max_iterations < 0) return QLAMD_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < 4; k++)
    if (params->leg_order[k] < 0 || params->leg_order[k] > 3) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  QL_ENTER(ctx, st);
  const size_t B = (size_t)batch;
  PoseParamsDev P;
  memcpy(P.hips, params->hip_in_base, sizeof(P.hips));
  P.com_weight = params->com_weight; P.tol = params->tolerance; P.max_iter = params->max_iterations;
  P.dummy_equality = params->dummy_equality;
  for (int k = 0; k < 4; k++) P.leg_order[k] = params->leg_order[k];

  PosePtrs s{in->stance, in->nominal_stance, in->support_polygon, in->center_of_mass, in->max_limb_length,
             in->pose, in->stance_mask, in->n_vertices};
  double *d_out = call.pose_out;
  int32_t *d_it = call.iterations, *d_st = call.status, *d_stage = call.stage;
  const double *d_min = call.min_len, *d_sfo = call.sfo;
  uint8_t *d_ok = call.ok;
  if (memory == QLAMD_MEM_HOST) {
    enum { kIn = 10 };
    const size_t sz[kIn] = {B * 96, B * 96, B * 64, in->center_of_mass ? B * 24 : 0, B * 32, in->pose ? B * 56 : 0,
                            in->stance_mask ? B * 4 : 0, in->n_vertices ? B * 4 : 0, call.min_len ? B * 32 : 0,
                            call.sfo ? B * 96 : 0};
    const void *src[kIn] = {in->stance, in->nominal_stance, in->support_polygon, in->center_of_mass,
                            in->max_limb_length, in->pose, in->stance_mask, in->n_vertices, call.min_len, call.sfo};
    size_t off[kIn + 5], total = 0;
    for (int k = 0; k < kIn; k++) { off[k] = total; total += align256(sz[k]); }
    const size_t osz[5] = {B * 56, B * 4, B * 4, B * 4, B};
    for (int k = 0; k < 5; k++) { off[kIn + k] = total; total += align256(osz[k]); }
    int rc = ensure_ws(ctx, total);
    if (rc != QLAMD_OK) return rc;
    char *w = (char *)ctx->ws;
    for (int k = 0; k < kIn; k++)
      if (sz[k] && hipMemcpyAsync(w + off[k], src[k], sz[k], hipMemcpyHostToDevice, st) != hipSuccess)
        return QLAMD_ERR_HIP;
    s = PosePtrs{(const double *)(w + off[0]), (const double *)(w + off[1]), (const double *)(w + off[2]),
                 in->center_of_mass ? (const double *)(w + off[3]) : nullptr, (const double *)(w + off[4]),
                 in->pose ? (const double *)(w + off[5]) : nullptr,
                 in->stance_mask ? (const uint8_t *)(w + off[6]) : nullptr,
                 in->n_vertices ? (const int32_t *)(w + off[7]) : nullptr};
    d_min = call.min_len ? (const double *)(w + off[8]) : nullptr;
    d_sfo = call.sfo ? (const double *)(w + off[9]) : nullptr;
    d_out = (double *)(w + off[kIn]);
    d_it = (int32_t *)(w + off[kIn + 1]);
    d_st = (int32_t *)(w + off[kIn + 2]);
    d_stage = (int32_t *)(w + off[kIn + 3]);
    d_ok = (uint8_t *)(w + off[kIn + 4]);
  }
  const unsigned rgrid = (unsigned)((batch + coop::kPoseCoopRows - 1) / coop::kPoseCoopRows); // 4 problems per wavefront
  switch (mode) {
    case kPoseSqp:
      hipLaunchKernelGGL(pose_sqp_coop_kernel, dim3(rgrid), dim3(64), 0, st, P, s, batch, d_out, d_it, d_st);
      break;
    case kPoseQp:
      hipLaunchKernelGGL(pose_qp_coop_kernel, dim3(rgrid), dim3(64), 0, st, P, s, batch, d_out, d_st);
      break;
    case kPoseCheck:
      hipLaunchKernelGGL(pose_check_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, P, s, d_min,
                         call.leg_tol, batch, d_ok);
      break;
    case kPoseGeometric:
      hipLaunchKernelGGL(pose_geometric_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, P, s, d_sfo, batch, d_out);
      break;
    case kPoseBaseAuto:
      hipLaunchKernelGGL(base_auto_coop_kernel, dim3(rgrid), dim3(64), 0, st, P, s, d_sfo, d_min, call.leg_tol, batch, d_out,
                         d_stage, d_it, d_st);
      break;
  }
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  if (memory == QLAMD_MEM_HOST) {
    const auto back = [&](void *dst, const void *srcp, size_t n) {
      return !dst || hipMemcpyAsync(dst, srcp, n, hipMemcpyDeviceToHost, st) == hipSuccess;
    };
    bool fine = true;
    if (mode != kPoseCheck) fine = fine && back(call.pose_out, d_out, B * 56);
    if (mode == kPoseSqp || mode == kPoseBaseAuto) fine = fine && back(call.iterations, d_it, B * 4);
    if (has_status) fine = fine && back(call.status, d_st, B * 4);
    if (mode == kPoseBaseAuto) fine = fine && back(call.stage, d_stage, B * 4);
    if (mode == kPoseCheck) fine = fine && back(call.ok, d_ok, B);
    if (!fine || hipStreamSynchronize(st) != hipSuccess) return QLAMD_ERR_HIP;
  }
  return QLAMD_OK;
}

int qlamd_pose_sqp_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in,
                         int64_t batch, double *pose_out, int32_t *iterations, int32_t *status, int memory,
                         void *stream) {
  PoseCall c{kPoseSqp};
  c.pose_out = pose_out; c.iterations = iterations; c.status = status;
  return pose_impl(c, ctx, params, in, batch, memory, stream);
}

int qlamd_pose_qp_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in, int64_t batch,
                        double *pose_out, int32_t *status, int memory, void *stream) {
  PoseCall c{kPoseQp};
  c.pose_out = pose_out; c.status = status;
  return pose_impl(c, ctx, params, in, batch, memory, stream);
}

int qlamd_pose_check_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in,
                           const double *min_limb_length, double leg_length_tolerance, int64_t batch, uint8_t *ok,
                           int memory, void *stream) {
  PoseCall c{kPoseCheck};
  c.ok = ok; c.min_len = min_limb_length; c.leg_tol = leg_length_tolerance;
  return pose_impl(c, ctx, params, in, batch, memory, stream);
}

int qlamd_pose_geometric_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in,
                               const double *stance_for_orientation, int64_t batch, double *pose_out, int memory,
                               void *stream) {
  PoseCall c{kPoseGeometric};
  c.pose_out = pose_out; c.sfo = stance_for_orientation;
  return pose_impl(c, ctx, params, in, batch, memory, stream);
}

int qlamd_base_auto_optimize_pose_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in,
                                        const double *stance_for_orientation, const double *min_limb_length,
                                        double leg_length_tolerance, int64_t batch, double *pose_out, int32_t *stage,
                                        int32_t *iterations, int32_t *status, int memory, void *stream) {
  PoseCall c{kPoseBaseAuto};
  c.pose_out = pose_out; c.stage = stage; c.iterations = iterations; c.status = status;
  c.sfo = stance_for_orientation; c.min_len = min_limb_length; c.leg_tol = leg_length_tolerance;
  return pose_impl(c, ctx, params, in, batch, memory, stream);
}

int qlamd_qp_solve_batch(qlamd_context *ctx, int n, int p, int m, const double *G, const double *g0,
                         const double *CE, const double *ce0, const double *CI, const double *ci0,
                         int64_t batch, double *x, double *objective, int32_t *status, int memory,
                         void *stream) {
  if (!ctx || batch < 0 || !G || !g0 || !x || !status) return QLAMD_ERR_INVALID_ARGUMENT;
  if (n < 1 || n > 12 || p < 0 || p > 2 || m < 0 || m > 48) return QLAMD_ERR_INVALID_ARGUMENT;
  if ((p > 0 && (!CE || !ce0)) || (m > 0 && (!CI || !ci0))) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  QL_ENTER(ctx, st);
  PlacePtrs pp;
  qlamd_placement pl;
  { const int rc = take_placement(ctx, memory, batch, &pp, &pl); if (rc != QLAMD_OK) return rc; }
  if (pp.prev_working_set || pp.working_set) return QLAMD_ERR_INVALID_ARGUMENT; // (the dense entries start cold)
  const size_t B = (size_t)batch;
  const double *dG = G, *dg0 = g0, *dCE = CE, *dce0 = ce0, *dCI = CI, *dci0 = ci0;
  double *dx = x, *dobj = objective;
  int32_t *dst = status;
  if (memory == QLAMD_MEM_HOST) {
    const size_t sz[6] = {B * n * n * 8, B * n * 8, B * n * p * 8, B * p * 8, B * n * m * 8, B * m * 8};
    const void *src[6] = {G, g0, CE, ce0, CI, ci0};
    size_t off[9], total = 0;
    for (int k = 0; k < 6; k++) { off[k] = total; total += align256(sz[k]); }
    off[6] = total; total += align256(B * n * 8);
    off[7] = total; total += align256(B * 8);
    off[8] = total; total += align256(B * 4);
    int rc = ensure_ws(ctx, total);
    if (rc != QLAMD_OK) return rc;
    char *w = (char *)ctx->ws;
    for (int k = 0; k < 6; k++)
      if (sz[k] && hipMemcpyAsync(w + off[k], src[k], sz[k], hipMemcpyHostToDevice, st) != hipSuccess)
        return QLAMD_ERR_HIP;
    dG = (const double *)(w + off[0]); dg0 = (const double *)(w + off[1]); dCE = (const double *)(w + off[2]);
    dce0 = (const double *)(w + off[3]); dCI = (const double *)(w + off[4]); dci0 = (const double *)(w + off[5]);
    dx = (double *)(w + off[6]); dobj = objective ? (double *)(w + off[7]) : nullptr; dst = (int32_t *)(w + off[8]);
  }
  {
    const unsigned cgrid = (unsigned)((batch + coop::kQpCoopRows - 1) / coop::kQpCoopRows);
    auto launch = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(cgrid), dim3(64), 0, st, n, p, m, dG, dg0, dCE, dce0, dCI, dci0, batch, dx, dobj, dst, pp);
    };
    if (m > 24) {
      if (n <= 6) launch(qp_coop_kernel<6, 3>); else launch(qp_coop_kernel<12, 3>);
    } else {
      if (n <= 6) launch(qp_coop_kernel<6, 2>); else launch(qp_coop_kernel<12, 2>);
    }
  }
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  { const int rc = finish_placement(ctx, pl, batch, st); if (rc != QLAMD_OK) return rc; }
  if (memory == QLAMD_MEM_HOST) {
    if (hipMemcpyAsync(x, dx, B * n * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (objective && hipMemcpyAsync(objective, dobj, B * 8, hipMemcpyDeviceToHost, st) != hipSuccess)
      return QLAMD_ERR_HIP;
    if (hipMemcpyAsync(status, dst, B * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return QLAMD_ERR_HIP;
  }
  return QLAMD_OK;
}

int qlamd_weighted_lsq_qp_batch(qlamd_context *ctx, int n, int k, int p, int m, const double *A, const double *S,
                                const double *b, const double *W, const double *C, const double *c, const double *D,
                                const double *d, const double *f, int64_t batch, double *x, int32_t *status, int memory,
                                void *stream) {
  if (!ctx || batch < 0 || !A || !S || !b || !W || !x || !status) return QLAMD_ERR_INVALID_ARGUMENT;
  if (n < 1 || n > 12 || k < 1 || k > 12 || p < 0 || p > 12 || m < 0 || m > 24) return QLAMD_ERR_INVALID_ARGUMENT;
  if ((p > 0 && (!C || !c)) || (m > 0 && (!D || !d || !f))) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  QL_ENTER(ctx, st);
  PlacePtrs pp;
  qlamd_placement pl;
  { const int rc = take_placement(ctx, memory, batch, &pp, &pl); if (rc != QLAMD_OK) return rc; }
  if (pp.prev_working_set || pp.working_set) return QLAMD_ERR_INVALID_ARGUMENT; // (the dense entries start cold)
  const size_t B = (size_t)batch;
  const double *dA = A, *dS = S, *db = b, *dW = W, *dC = p ? C : nullptr, *dc = p ? c : nullptr, *dD = m ? D : nullptr,
               *dd = m ? d : nullptr, *df = m ? f : nullptr;
  double *dx = x;
  int32_t *dst = status;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    const int iA = sg.add(A, B * k * n * 8, true, false), iS = sg.add(S, B * k * 8, true, false), ib = sg.add(b, B * k * 8, true, false);
    const int iW = sg.add(W, B * n * 8, true, false), iC = sg.add(dC, B * p * n * 8, true, false), ic = sg.add(dc, B * p * 8, true, false);
    const int iD = sg.add(dD, B * m * n * 8, true, false), id = sg.add(dd, B * m * 8, true, false), iF = sg.add(df, B * m * 8, true, false);
    const int ox = sg.add(x, B * n * 8, false, true), os = sg.add(status, B * 4, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    dA = sg.dev<const double>(iA); dS = sg.dev<const double>(iS); db = sg.dev<const double>(ib); dW = sg.dev<const double>(iW);
    dC = sg.dev<const double>(iC); dc = sg.dev<const double>(ic); dD = sg.dev<const double>(iD); dd = sg.dev<const double>(id);
    df = sg.dev<const double>(iF); dx = sg.dev<double>(ox); dst = sg.dev<int32_t>(os);
  }
  const unsigned grid = (unsigned)((batch + coop::kQpCoopRows - 1) / coop::kQpCoopRows);
  if (n <= 6)
    hipLaunchKernelGGL(weighted_lsq_qp_kernel<6>, dim3(grid), dim3(64), 0, st, n, k, p, m, dA, dS, db, dW, dC, dc, dD, dd, df, batch, dx, dst, pp);
  else
    hipLaunchKernelGGL(weighted_lsq_qp_kernel<12>, dim3(grid), dim3(64), 0, st, n, k, p, m, dA, dS, db, dW, dC, dc, dD, dd, df, batch, dx, dst, pp);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  { const int rc = finish_placement(ctx, pl, batch, st); if (rc != QLAMD_OK) return rc; }
  return memory == QLAMD_MEM_HOST ? sg.finish(st) : QLAMD_OK;
}

} // extern "C"

QLAMD_STAMPS_ACCESSOR(qlamd_debug_stamps_pose)
