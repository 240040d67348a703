// HIP kernels (gfx950) of the floating-base dynamics and the whole-body QP and their
// part of the C-ABI of include/qlamd.h.
#include "wholebody_coop.hpp"
#include "context.hpp"

using namespace qlamd;
using namespace qlamd::rt;

namespace {

// ---- whole-body (floating-base) dynamics and QP, csrc/wholebody_coop.hpp ----------
struct WbPtrs {
  const double *q, *qd, *quat, *linvel, *angvel, *a_des, *qdd;
  const uint8_t *stance;
  const double *normals;
};

// What every whole-body kernel loads per lane, all issued before the first use.
struct WbLaneIn {
  double quat[4], linvel[3], angvel[3];
  double qj, qdj;
  __device__ __forceinline__ void load(const WbPtrs &s, int64_t i, int joint) {
    const double2 *a2 = reinterpret_cast<const double2 *>(s.quat + 4 * i);
    double2 v = a2[0]; quat[0] = v.x; quat[1] = v.y;
    v = a2[1]; quat[2] = v.x; quat[3] = v.y;
#pragma unroll
    for (int k = 0; k < 3; k++) { linvel[k] = s.linvel[3 * i + k]; angvel[k] = s.angvel[3 * i + k]; }
    qj = s.q[12 * i + joint];
    qdj = s.qd[12 * i + joint];
  }
};

// n doubles from LDS to global memory by the 64 lanes of the block, 16 bytes per lane and instruction (both sides are
// 16-byte aligned: every per-robot record here is a multiple of 16 bytes and n is even)
__device__ __forceinline__ void copy_out(double *__restrict__ dst, const double *src, int n) {
  const double2 *s2 = reinterpret_cast<const double2 *>(src);
  double2 *d2 = reinterpret_cast<double2 *>(dst);
  for (int e = threadIdx.x; e < (n >> 1); e += 64) d2[e] = s2[e];
}

// M [B][18][18], h [B][18], Jc [B][12][18] (any of them may be NULL): staged per robot in LDS, written out coalesced.
// kM: the composite-rigid-body pass and M; kHJ: the Newton-Euler pass, h and Jc (a caller that wants only one of the two
// does not pay for the other).
template <bool kM, bool kHJ>
__global__ __launch_bounds__(64) void wholebody_dynamics_kernel(const DeviceParams *__restrict__ Pp, const coop::WbParamsDev W,
                                                                const WbPtrs s, int64_t B, double *__restrict__ Mo,
                                                                double *__restrict__ ho, double *__restrict__ Jo) {
  using namespace coop;
  __shared__ double tab[4 * kTabPerLeg];
  __shared__ double outb[4 * kWbStage]; // staged twice: M, then h and Jc
  const DeviceParams &P = *Pp;
  TabStage ts;
  ts.issue(P);
  const int row = threadIdx.x >> 4, lr = threadIdx.x & 15, leg = lr >> 2, c = lr & 3;
  const int64_t i0 = (int64_t)blockIdx.x * 4 + row;
  const int64_t i = i0 < B ? i0 : B - 1;
  WbLaneIn in;
  in.load(s, i, 3 * leg + (c < 3 ? c : 2));
  ts.commit(tab);
  double *ob = outb + kWbStage * row;
  const double2 zero2 = {0.0, 0.0};
  if (kM && Mo) {
    for (int e = lr; e < kWbStage / 2; e += 16) reinterpret_cast<double2 *>(ob)[e] = zero2;
  } else {
    for (int e = threadIdx.x; e < 2 * (18 + 216); e += 64) reinterpret_cast<double2 *>(outb)[e] = zero2;
  }

  double Rm[9], gB[3];
  quat_to_matrix(in.quat, Rm);
  const double gW[3] = {0.0, 0.0, -W.grav};
  irot(Rm, gW, gB);
  double vB[3];
  irot(Rm, in.linvel, vB);
  double sj, cj;
  sincos_reduced(in.qj, sj, cj);
  WbLink L;
  wb_link(CoopTab{tab + kTabPerLeg * leg}, c, sj, cj, L);
  const double V0[6] = {in.angvel[0], in.angvel[1], in.angvel[2], vB[0], vB[1], vB[2]};
  const double A0[6] = {0.0, 0.0, 0.0, -gB[0], -gB[1], -gB[2]};
  WbInertia T{};
  double Fcol[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, Mleg[3] = {0.0, 0.0, 0.0};
  if constexpr (kM) wb_crba(W, L, c, T, Fcol, Mleg);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);

  const int64_t r0 = (int64_t)blockIdx.x * 4;
  const int nrob = (int)((B - r0) < 4 ? (B - r0) : 4);
  // ---- pass 1: the mass matrix.  Base block, interface order [linear ; angular]:  [[m 1, -[h]x], [[h]x, I]]
  if (kM && Mo) {
    static_for<36>([&](auto E) {
      constexpr int e = E, a = e / 6, b = e % 6;
      double val = 0.0;
      if constexpr (a < 3 && b < 3) {
        val = a == b ? T.m : 0.0;
      } else if constexpr (a >= 3 && b >= 3) {
        constexpr int r = a - 3, q = b - 3, lo = r < q ? r : q, hi = r < q ? q : r;
        val = T.I[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
      } else {
        // [h]x entry (r, q) = -+ h[3 - r - q]; the upper-right block is -[h]x, the lower-left +[h]x
        constexpr int r = a < 3 ? a : a - 3, q = b < 3 ? b : b - 3;
        if constexpr (r != q) {
          constexpr bool neg = ((q - r + 3) % 3 == 1) != (a < 3);
          val = neg ? -T.h[3 - r - q] : T.h[3 - r - q];
        }
      }
      ob[kWbM + 18 * a + b] = val; // replicated value: every lane of the row stores it
    });
    if (c < 3) {
      const int j = 6 + 3 * leg + c;
#pragma unroll
      for (int a = 0; a < 3; a++) { // my column / row of the base block: [force ; moment]
        ob[kWbM + 18 * a + j] = Fcol[3 + a]; ob[kWbM + 18 * j + a] = Fcol[3 + a];
        ob[kWbM + 18 * (3 + a) + j] = Fcol[a]; ob[kWbM + 18 * j + 3 + a] = Fcol[a];
        ob[kWbM + 18 * j + 6 + 3 * leg + a] = Mleg[a];
      }
    }
    __syncthreads();
    copy_out(Mo + r0 * 324, outb, 324 * nrob); // kWbStage == 324: robots are contiguous
    __syncthreads();
    if (kHJ && (ho || Jo))
      for (int e = threadIdx.x; e < 2 * (18 + 216); e += 64) reinterpret_cast<double2 *>(outb)[e] = zero2;
  }
  // ---- pass 2: bias forces and the contact Jacobian; the block holds h of its 4 robots, then Jc of its 4 robots, so
  //      that both go out as plain contiguous copies.  The Newton-Euler pass runs HERE, after the mass matrix has been
  //      handed to the memory system: its arithmetic overlaps the drain of those stores.
  if (kHJ && (ho || Jo)) {
    double tau = 0.0, gb[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    wb_inverse_dynamics(W, L, c, V0, A0, c < 3 ? in.qdj : 0.0, 0.0, tau, gb);
    double *hb = outb + 18 * row, *jb = outb + 4 * 18 + 216 * row;
    __syncthreads(); // the block-wide zero fill is complete
    static_for<6>([&](auto E) { constexpr int e = E; hb[e] = gb[e]; });
    if (c < 3) {
      const int j = 6 + 3 * leg + c;
      hb[j] = tau;
      double d[3] = {L.pf[0] - L.p[0], L.pf[1] - L.p[1], L.pf[2] - L.p[2]}, col[3];
      cross3(L.z, d, col);
#pragma unroll
      for (int a = 0; a < 3; a++) jb[18 * (3 * leg + a) + j] = col[a];
    } else {
      // the foot lane writes [1 , -[r]x] of its leg's three rows
      double *jr = jb + 18 * 3 * leg;
      jr[0] = 1.0; jr[18 + 1] = 1.0; jr[36 + 2] = 1.0;
      jr[4] = L.pf[2]; jr[5] = -L.pf[1];            // -[r]x
      jr[18 + 3] = -L.pf[2]; jr[18 + 5] = L.pf[0];
      jr[36 + 3] = L.pf[1]; jr[36 + 4] = -L.pf[0];
    }
    __syncthreads();
    if (ho) copy_out(ho + r0 * 18, outb, 18 * nrob);
    if (Jo) copy_out(Jo + r0 * 216, outb + 4 * 18, 216 * nrob);
  }
}

// One whole-body control step per robot: inverse dynamics for the desired accelerations -> force/torque QP over the
// stance legs (12 force variables, torques eliminated through the joint rows; 11 inequality rows per stance leg)
// -> joint efforts.  Quad lanes (leg, body) do the dynamics, then lane i < 12 carries variable i = 3 leg + c and lane
// j the inequalities j, j + 16, j + 32 with id = 11 leg + t: t = 0 minimal normal force, 1..4 friction pyramid,
// 5 + 2k (+1) upper (lower) torque bound of joint k.
template <bool kPerLeg>
__global__ __launch_bounds__(64) void wholebody_solve_kernel(const DeviceParams *__restrict__ Pp, const coop::WbParamsDev W,
                                                             const WbPtrs s, int64_t B, double *__restrict__ tau_out,
                                                             double *__restrict__ grf_out, int32_t *__restrict__ status_out) {
  using namespace coop;
  typedef QpCoopLds<12, 3> L3;
  __shared__ double tab[4 * kTabPerLeg];
  __shared__ double xch[4 * kWxDoubles];
  __shared__ double qpl[4 * L3::kTotal];
  const DeviceParams &P = *Pp;
  TabStage ts;
  ts.issue(P);
  const int row = threadIdx.x >> 4, lr = threadIdx.x & 15, leg = lr >> 2, c = lr & 3;
  const int64_t i0 = (int64_t)blockIdx.x * 4 + row;
  const bool live = i0 < B;
  const int64_t i = live ? i0 : B - 1;
  const int jq = 3 * leg + (c < 3 ? c : 2);
  WbLaneIn in;
  in.load(s, i, jq);
  double ades[6];
#pragma unroll
  for (int k = 0; k < 6; k++) ades[k] = s.a_des[6 * i + k];
  const double qdd_raw = (s.qdd ? s.qdd : s.qd)[12 * i + jq];
  const uint32_t sm = *reinterpret_cast<const uint32_t *>(s.stance + 4 * i);
  double nWl[3] = {0.0, 0.0, 1.0};
  if (kPerLeg) { nWl[0] = s.normals[12 * i + 3 * leg]; nWl[1] = s.normals[12 * i + 3 * leg + 1]; nWl[2] = s.normals[12 * i + 3 * leg + 2]; }
  ts.commit(tab);
  const unsigned stance = live ? (((sm & 0xFFu) ? 1u : 0u) | ((sm & 0xFF00u) ? 2u : 0u) | ((sm & 0xFF0000u) ? 4u : 0u) |
                                  ((sm & 0xFF000000u) ? 8u : 0u))
                               : 0u;
  const int nS = __popc(stance);
  double *xb = xch + kWxDoubles * row;

  // ---- dynamics on the quad lanes
  double Rm[9], gB[3], vB[3];
  quat_to_matrix(in.quat, Rm);
  const double gW[3] = {0.0, 0.0, -W.grav};
  irot(Rm, gW, gB);
  irot(Rm, in.linvel, vB);
  double sj, cj;
  sincos_reduced(in.qj, sj, cj);
  WbLink L;
  wb_link(CoopTab{tab + kTabPerLeg * leg}, c, sj, cj, L);
  const double V0[6] = {in.angvel[0], in.angvel[1], in.angvel[2], vB[0], vB[1], vB[2]};
  const double A0[6] = {ades[3], ades[4], ades[5], ades[0] - gB[0], ades[1] - gB[1], ades[2] - gB[2]};
  double tau0, gb[6];
  wb_inverse_dynamics(W, L, c, V0, A0, c < 3 ? in.qdj : 0.0, (c < 3 && s.qdd) ? qdd_raw : 0.0, tau0, gb);
  // friction pyramid of my leg (ContactForceDistribution.cpp:254-336, as in balance_coop.hpp)
  double nb[3], t1[3], t2[3];
  {
    const double ey[3] = {0.0, 1.0, 0.0}, ez[3] = {0.0, 0.0, 1.0};
    double yB[3], nW[3];
    irot(Rm, ey, yB);
    if (kPerLeg) { nW[0] = nWl[0]; nW[1] = nWl[1]; nW[2] = nWl[2]; }
    else rot(Rm, ez, nW);
    irot(Rm, nW, nb);
    cross3(nb, yB, t1);
    double nn = rsqrt_nr(dot3(t1, t1));
    t1[0] *= nn; t1[1] *= nn; t1[2] *= nn;
    cross3(nb, t1, t2);
    nn = rsqrt_nr(dot3(t2, t2));
    t2[0] *= nn; t2[1] *= nn; t2[2] *= nn;
  }
  // ---- exchange through LDS: quad layout -> variable / constraint layout
  if (c < 3) {
    xb[kWxTau0 + 3 * leg + c] = tau0;
    const double d[3] = {L.pf[0] - L.p[0], L.pf[1] - L.p[1], L.pf[2] - L.p[2]};
    double col[3];
    cross3(L.z, d, col);
#pragma unroll
    for (int a = 0; a < 3; a++) xb[kWxJ + 9 * leg + 3 * a + c] = col[a];
  } else {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      xb[kWxR + 3 * leg + a] = L.pf[a];
      xb[kWxN + 9 * leg + a] = nb[a]; xb[kWxN + 9 * leg + 3 + a] = t1[a]; xb[kWxN + 9 * leg + 6 + a] = t2[a];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);

  // ---- QP data.  Variable lane i = 3 vl + vc.
  const int vi = lr < 12 ? lr : 0, vl = (vi * 11) >> 5, vc = vi - 3 * vl; // vi / 3 for vi < 12
  const bool von = lr < 12 && ((stance >> vl) & 1u);
  double rl[3], Jl[9], tl[3];
#pragma unroll
  for (int a = 0; a < 3; a++) { rl[a] = xb[kWxR + 3 * vl + a]; tl[a] = xb[kWxTau0 + 3 * vl + a]; }
#pragma unroll
  for (int a = 0; a < 9; a++) Jl[a] = xb[kWxJ + 9 * vl + a];
  // column vc of [r_l]x, weighted
  const double av[3] = {sel(vc == 1, -rl[2], sel(vc == 2, rl[1], 0.0)), sel(vc == 0, rl[2], sel(vc == 2, -rl[0], 0.0)),
                        sel(vc == 0, -rl[1], sel(vc == 1, rl[0], 0.0))};
  const double sa[3] = {P.S[3] * av[0], P.S[4] * av[1], P.S[5] * av[2]};
  const double Sfc = pick3(P.S, vc);
  double jr[3]; // row vc of J_leg: J[vc][k]
#pragma unroll
  for (int k = 0; k < 3; k++) jr[k] = sel(vc == 0, Jl[k], sel(vc == 1, Jl[3 + k], Jl[6 + k]));
  double Gm[12];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const bool both = von && ((stance >> m) & 1u);
    const double xp = xb[kWxR + 3 * m], yp = xb[kWxR + 3 * m + 1], zp = xb[kWxR + 3 * m + 2];
    const double e0 = sa[1] * zp - sa[2] * yp;
    const double e1 = -sa[0] * zp + sa[2] * xp;
    const double e2 = sa[0] * yp - sa[1] * xp;
    // torque regulariser w_tau J J' on my own leg's block
    const bool own = m == vl;
    const double jj0 = jr[0] * Jl[0] + jr[1] * Jl[1] + jr[2] * Jl[2];
    const double jj1 = jr[0] * Jl[3] + jr[1] * Jl[4] + jr[2] * Jl[5];
    const double jj2 = jr[0] * Jl[6] + jr[1] * Jl[7] + jr[2] * Jl[8];
    Gm[3 * m + 0] = both ? e0 + (vc == 0 ? Sfc : 0.0) + (own ? W.w_tau * jj0 : 0.0) : 0.0;
    Gm[3 * m + 1] = both ? e1 + (vc == 1 ? Sfc : 0.0) + (own ? W.w_tau * jj1 : 0.0) : 0.0;
    Gm[3 * m + 2] = both ? e2 + (vc == 2 ? Sfc : 0.0) + (own ? W.w_tau * jj2 : 0.0) : 0.0;
  }
#pragma unroll
  for (int j = 0; j < 12; j++) Gm[j] += (lr == j) ? (von ? P.w_reg : 1.0) : 0.0; // identity row for a swing-leg variable
  const double ST[3] = {P.S[3] * gb[3], P.S[4] * gb[4], P.S[5] * gb[5]};
  const double Fc = pick3(gb, vc);
  const double g0 = von ? -(Sfc * Fc + (av[0] * ST[0] + av[1] * ST[1] + av[2] * ST[2]) +
                            W.w_tau * (jr[0] * tl[0] + jr[1] * tl[1] + jr[2] * tl[2]))
                        : 0.0;
  // my three inequalities
  double a[3][12], b[3];
  bool v[3];
#pragma unroll
  for (int sidx = 0; sidx < 3; sidx++) {
    const int id = lr + 16 * sidx;
    const int cl = (id * 47) >> 9;                 // id / 11 for id < 48
    const int t = id - 11 * cl;
    const int cll = cl < 4 ? cl : 3;
    v[sidx] = id < 44 && ((stance >> cll) & 1u);
    const double *nrm = xb + kWxN + 9 * cll;
    const int k = t >= 5 ? ((t - 5) >> 1) : 0;
    const bool lower = t >= 5 && ((t - 5) & 1);
    double nv[3];
#pragma unroll
    for (int e = 0; e < 3; e++) {
      const double fr = P.mu * nrm[e] + ((t == 1) ? nrm[3 + e] : (t == 2) ? -nrm[3 + e] : (t == 3) ? nrm[6 + e] : -nrm[6 + e]);
      const double jc = xb[kWxJ + 9 * cll + 3 * e + k];
      nv[e] = t == 0 ? nrm[e] : (t < 5 ? fr : (lower ? -jc : jc));
    }
    const double t0k = xb[kWxTau0 + 3 * cll + k];
    b[sidx] = !v[sidx] ? 0.0 : (t == 0 ? -P.f_min : (t < 5 ? 0.0 : (lower ? W.tau_max + t0k : W.tau_max - t0k)));
#pragma unroll
    for (int j = 0; j < 12; j++) a[sidx][j] = (v[sidx] && (j / 3) == cll) ? nv[j % 3] : 0.0;
  }
  double x, fobj;
  const int st = qp_coop_impl<12, 3>(Gm, g0, 12, 3 * nS, 44, false, 0.0, 0.0, a, b, v, !live || nS == 0,
                                      qpl + L3::kTotal * row, x, fobj);
  (void)fobj;
  // ---- joint efforts: tau = tau0 - J_leg' f on the stance legs, tau0 elsewhere
  const bool ok = st == kStatusOk;
  if (lr < 12) xb[kWxX + lr] = von && ok ? x : 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if (lr < 12 && live) {
    const double f0 = xb[kWxX + 3 * vl], f1 = xb[kWxX + 3 * vl + 1], f2 = xb[kWxX + 3 * vl + 2];
    // joint vc of leg vl: column vc of J_leg
    const double tq = pick3(tl, vc) - (sel(vc == 0, Jl[0], sel(vc == 1, Jl[1], Jl[2])) * f0 +
                                       sel(vc == 0, Jl[3], sel(vc == 1, Jl[4], Jl[5])) * f1 +
                                       sel(vc == 0, Jl[6], sel(vc == 1, Jl[7], Jl[8])) * f2);
    tau_out[12 * i + lr] = ok ? tq : 0.0;
    if (grf_out) grf_out[12 * i + lr] = xb[kWxX + lr];
  }
  if (lr == 0 && live) status_out[i] = st;
}

} // namespace

extern "C" {

void qlamd_wholebody_default_params(qlamd_wholebody_params *p) {
  if (!p) return;
  p->torque_weight = 1e-3;
  p->torque_limit = 300.0; // the clamp of ros_balance_controller.cpp:451-454, here a constraint
  p->gravity = 9.81;       // RBDL's default, what the reference's swing-leg model uses (model_test_header.cpp:229-244)
}

static coop::WbParamsDev wb_params_of(const qlamd_context *ctx, double w_tau, double tau_max, double gravity) {
  coop::WbParamsDev W;
  W.base_m = ctx->base_m;
  for (int a = 0; a < 3; a++) W.base_h[a] = ctx->base_h[a];
  for (int a = 0; a < 6; a++) W.base_I[a] = ctx->base_I[a];
  W.w_tau = w_tau; W.tau_max = tau_max; W.grav = gravity;
  return W;
}

int qlamd_wholebody_dynamics_batch(qlamd_context *ctx, const qlamd_wholebody_batch *in, double gravity, int64_t batch,
                                   double *mass_matrix, double *nonlinear_effects, double *contact_jacobian,
                                   int memory, void *stream) {
  if (!ctx || !in || batch < 0 || !(mass_matrix || nonlinear_effects || contact_jacobian)) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!in->joint_position || !in->joint_velocity || !in->base_orientation || !in->base_linear_velocity ||
      !in->base_angular_velocity)
    return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  QL_ENTER(ctx, st);
  const size_t B = (size_t)batch;
  WbPtrs s{in->joint_position, in->joint_velocity, in->base_orientation, in->base_linear_velocity,
           in->base_angular_velocity, nullptr, nullptr, nullptr, nullptr};
  double *dM = mass_matrix, *dh = nonlinear_effects, *dJ = contact_jacobian;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    sg.add(in->joint_position, B * 96, true, false);
    sg.add(in->joint_velocity, B * 96, true, false);
    sg.add(in->base_orientation, B * 32, true, false);
    sg.add(in->base_linear_velocity, B * 24, true, false);
    sg.add(in->base_angular_velocity, B * 24, true, false);
    sg.add(mass_matrix, B * 324 * 8, false, true);
    sg.add(nonlinear_effects, B * 18 * 8, false, true);
    sg.add(contact_jacobian, B * 216 * 8, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    s.q = sg.dev<const double>(0); s.qd = sg.dev<const double>(1); s.quat = sg.dev<const double>(2);
    s.linvel = sg.dev<const double>(3); s.angvel = sg.dev<const double>(4);
    dM = sg.dev<double>(5); dh = sg.dev<double>(6); dJ = sg.dev<double>(7);
  }
  const coop::WbParamsDev W = wb_params_of(ctx, 0.0, 0.0, gravity);
  const dim3 grid((unsigned)((batch + 3) / 4));
  // One launch for everything.  Two launches (M; h and Jc) need 140 / 158 instead of 204 registers, i.e. three waves
  // per SIMD instead of two, but repeat the link kinematics: measured 9 % slower at 65 536 robots, 30 % at 4096
  // (QLAMD_WB_SPLIT=1 selects them, for measurement).
  const bool fused = !ctx->wb_split;
  if (fused && dM && (dh || dJ)) {
    hipLaunchKernelGGL((wholebody_dynamics_kernel<true, true>), grid, dim3(64), 0, st, ctx->d_params, W, s, batch, dM, dh, dJ);
  } else {
    if (dM)
      hipLaunchKernelGGL((wholebody_dynamics_kernel<true, false>), grid, dim3(64), 0, st, ctx->d_params, W, s, batch, dM,
                         (double *)nullptr, (double *)nullptr);
    if (dh || dJ)
      hipLaunchKernelGGL((wholebody_dynamics_kernel<false, true>), grid, dim3(64), 0, st, ctx->d_params, W, s, batch,
                         (double *)nullptr, dh, dJ);
  }
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

int qlamd_wholebody_solve_batch(qlamd_context *ctx, const qlamd_wholebody_params *params,
                                const qlamd_wholebody_batch *in, int64_t batch, double *joint_effort,
                                double *contact_force, int32_t *status, int memory, void *stream) {
  if (!ctx || !in || batch < 0 || !joint_effort || !status) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!params) return QLAMD_ERR_NOT_LOADED;
  if (!in->joint_position || !in->joint_velocity || !in->base_orientation || !in->base_linear_velocity ||
      !in->base_angular_velocity || !in->desired_base_acceleration || !in->support_leg)
    return QLAMD_ERR_INVALID_ARGUMENT;
  if (!(params->torque_weight > 0.0) || !(params->torque_limit > 0.0)) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  QL_ENTER(ctx, st);
  const size_t B = (size_t)batch;
  WbPtrs s{in->joint_position, in->joint_velocity, in->base_orientation, in->base_linear_velocity,
           in->base_angular_velocity, in->desired_base_acceleration, in->desired_joint_acceleration, in->support_leg,
           in->surface_normal};
  double *dtau = joint_effort, *dgrf = contact_force;
  int32_t *dst = status;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    sg.add(in->joint_position, B * 96, true, false);
    sg.add(in->joint_velocity, B * 96, true, false);
    sg.add(in->base_orientation, B * 32, true, false);
    sg.add(in->base_linear_velocity, B * 24, true, false);
    sg.add(in->base_angular_velocity, B * 24, true, false);
    sg.add(in->desired_base_acceleration, B * 48, true, false);
    sg.add(in->desired_joint_acceleration, B * 96, true, false);
    sg.add(in->support_leg, B * 4, true, false);
    sg.add(in->surface_normal, B * 96, true, false);
    sg.add(joint_effort, B * 96, false, true);
    sg.add(contact_force, B * 96, false, true);
    sg.add(status, B * 4, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    s = WbPtrs{sg.dev<const double>(0), sg.dev<const double>(1), sg.dev<const double>(2), sg.dev<const double>(3),
               sg.dev<const double>(4), sg.dev<const double>(5), sg.dev<const double>(6), sg.dev<const uint8_t>(7),
               sg.dev<const double>(8)};
    dtau = sg.dev<double>(9); dgrf = sg.dev<double>(10); dst = sg.dev<int32_t>(11);
  }
  const coop::WbParamsDev W = wb_params_of(ctx, params->torque_weight, params->torque_limit, params->gravity);
  const unsigned grid = (unsigned)((batch + 3) / 4);
  if (s.normals)
    hipLaunchKernelGGL(wholebody_solve_kernel<true>, dim3(grid), dim3(64), 0, st, ctx->d_params, W, s, batch, dtau, dgrf, dst);
  else
    hipLaunchKernelGGL(wholebody_solve_kernel<false>, dim3(grid), dim3(64), 0, st, ctx->d_params, W, s, batch, dtau, dgrf, dst);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

} // extern "C"
