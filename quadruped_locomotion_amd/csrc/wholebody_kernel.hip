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
// stance legs -> joint efforts.  The torques are eliminated through the joint rows (tau = tau0 - J_leg' f), which leaves
// 12 force variables and 11 inequality rows per stance leg (minimal normal force, friction pyramid, upper and lower
// bound of each joint torque), every one of them local to a leg: the QP runs on the quad layout the dynamics was
// computed in (lane 4 leg + c: body c of the leg = force component c = joint c), force_qp_coop.hpp with its torque rows
// -- no exchange through LDS, no general dense solver.
template <bool kPerLeg>
__global__ __launch_bounds__(64) void wholebody_solve_kernel(const DeviceParams *__restrict__ Pp, const coop::WbParamsDev W,
                                                             const WbPtrs s, int64_t B, double *__restrict__ tau_out,
                                                             double *__restrict__ grf_out, int32_t *__restrict__ status_out) {
  using namespace coop;
  __shared__ double tab[4 * kTabPerLeg];
  __shared__ double rows[4 * kCoopLdsDoubles];
  __shared__ double nrm[kForceQpNrmRowsTorque * 64];
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
  const bool comp = c < 3, on = ((stance >> leg) & 1u) != 0u, row_on = comp && on;

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
  wb_inverse_dynamics(W, L, c, V0, A0, comp ? in.qdj : 0.0, (comp && s.qdd) ? qdd_raw : 0.0, tau0, gb);
  tau0 = comp ? tau0 : 0.0;

  ForceQp Q;
  // friction pyramid of my leg (ContactForceDistribution.cpp:254-336, as in balance_coop.hpp)
  {
    const double ey[3] = {0.0, 1.0, 0.0}, ez[3] = {0.0, 0.0, 1.0};
    double yB[3], nW[3];
    irot(Rm, ey, yB);
    if (kPerLeg) { nW[0] = nWl[0]; nW[1] = nWl[1]; nW[2] = nWl[2]; }
    else rot(Rm, ez, nW);
    irot(Rm, nW, Q.nb);
    cross3(Q.nb, yB, Q.t1);
    double nn = rsqrt_nr(dot3(Q.t1, Q.t1));
    Q.t1[0] *= nn; Q.t1[1] *= nn; Q.t1[2] *= nn;
    cross3(Q.nb, Q.t1, Q.t2);
    nn = rsqrt_nr(dot3(Q.t2, Q.t2));
    Q.t2[0] *= nn; Q.t2[1] *= nn; Q.t2[2] *= nn;
    Q.myn = pick3(Q.nb, c); Q.myt1 = pick3(Q.t1, c); Q.myt2 = pick3(Q.t2, c);
  }
  // contact Jacobian of my leg: my lane's joint gives COLUMN c, J[a][c] = (z_c x (p_foot - p_c))[a]; row c is what the
  // lanes of the quad hold at index c
  {
    const double d[3] = {L.pf[0] - L.p[0], L.pf[1] - L.p[1], L.pf[2] - L.p[2]};
    double col[3];
    cross3(L.z, d, col);
#pragma unroll
    for (int a = 0; a < 3; a++) Q.jcol[a] = row_on ? col[a] : 0.0;
    static_for<3>([&](auto K) {
      constexpr int k = K;
      const double v[3] = {quad_bc<k>(Q.jcol[0]), quad_bc<k>(Q.jcol[1]), quad_bc<k>(Q.jcol[2])};
      Q.jrow[k] = pick3(v, c);
    });
  }
  const double jt0 = Q.jrow[0] * quad_bc<0>(tau0) + Q.jrow[1] * quad_bc<1>(tau0) + Q.jrow[2] * quad_bc<2>(tau0);
  const double foot = row_on ? pick3(L.pf, c) : 0.0;
  force_qp_objective(P.S, P.w_reg, foot, stance, row_on, gb, Q.jrow, W.w_tau * jt0, Q.Gm, Q.g0, W.w_tau);
  Q.mu = P.mu; Q.f_min = P.f_min;
  Q.on = on; Q.comp = comp; Q.nS = nS; Q.refine_passes = P.refine_passes;
  Q.tq_up = W.tau_max - tau0; Q.tq_lo = W.tau_max + tau0;
  double x = 0.0;
  const int st = force_qp_coop<true>(Q, rows + kCoopLdsDoubles * row, nrm, x);

  // ---- joint efforts: tau = tau0 - J_leg' f on the stance legs, tau0 elsewhere
  const bool ok = st == kStatusOk;
  const double f = (row_on && ok) ? x : 0.0;
  const double tq = tau0 - (Q.jcol[0] * quad_bc<0>(f) + Q.jcol[1] * quad_bc<1>(f) + Q.jcol[2] * quad_bc<2>(f));
  if (comp && live) {
    tau_out[12 * i + 3 * leg + c] = ok ? tq : 0.0;
    if (grf_out) grf_out[12 * i + 3 * leg + c] = f;
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
