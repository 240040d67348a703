// Lane-cooperative ("latency") form of the balance-controller step: 16 lanes per robot,
// 4 robots per wavefront.  Device-only (DPP cross-lane moves); included by balance_kernel.hip.
//
// Why: with a few thousand robots the chip runs ~1 wavefront per SIMD and a control step
// lasts as long as the slowest robot's SERIAL instruction stream.  Spreading one robot over a
// 16-lane DPP row turns every 12x12 product into 12 broadcasts + 12 FMAs per lane and every
// search over the 20 constraints into one DPP reduction.
//
// Lane r = 4*leg + c of a row: c = 0,1,2 carries component c of leg `leg` of every 12-vector
// (x, z, g0 ...) and row i = 3*leg + c of every 12x12 matrix (G, H); c = 3 is a spare lane.
// Lane k (0..11) also carries slot k of the active set: id_k, multiplier u_k, row k of N*.
// Per-robot scalars are replicated on the 16 lanes; control flow is uniform inside a row.
//
// QP method: Goldfarb-Idnani with the operators of the original paper kept EXPLICITLY,
//   H  = G^-1 - G^-1 N (N'G^-1 N)^-1 N'G^-1      (12x12, row per lane)
//   N* = (N'G^-1 N)^-1 N'G^-1                     (q x 12, row per slot lane)
// so that z = H n_p and r = N* n_p are one broadcast pass, and adding / dropping a
// constraint is a rank-one update:
//   add n+:  H -= z z'/d,  N* <- [N* - r z'/d ; z'/d],          d = z'n+
//   drop k:  H += n~ n~'/e, N* <- rows!=k of (N* - (N* G n~) n~'/e), n~ = row k of N*, e = n~'G n~
// Same pivot rule, step lengths and termination test as QuadProg++ (QuadProg++.cc:216-445).
// The explicit operators drift by ~1e-9 per update; a final refinement on the known working
// set (constraint residual through N*', reduced gradient through H) removes the drift.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "balance_core.hpp"

namespace qlamd {
namespace coop {

// Diagnostic build only (-DQLAMD_STAMPS): s_memtime at segment boundaries of wave 0, read back
// through qlamd_debug_stamps.  Never compiled into the shipped library.
#ifdef QLAMD_STAMPS
__device__ unsigned long long g_stamps[64];
#define QL_STAMP(k)                                                                         \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0) ::qlamd::coop::g_stamps[k] = t_;                              \
  } while (0)
// segment accumulators inside the active-set loop: QL_SEG(k) adds the time since the previous QL_SEG to slot k
#define QL_SEG_DECL unsigned long long ql_last_ = 0, ql_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define QL_SEG_START                                                                        \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ql_last_)::"memory");     \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#define QL_SEG(k)                                                                           \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    ql_acc_[k] += t_ - ql_last_;                                                            \
    ql_last_ = t_;                                                                          \
  } while (0)
#define QL_SEG_STORE                                                                        \
  do {                                                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0)                                                \
      for (int k_ = 0; k_ < 8; k_++) ::qlamd::coop::g_stamps[16 + k_] = ql_acc_[k_];        \
  } while (0)
#else
#define QL_STAMP(k)
#define QL_SEG_DECL
#define QL_SEG_START
#define QL_SEG(k)
#define QL_SEG_STORE
#endif

template <int CTRL>
__device__ __forceinline__ double dpp(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  // mov_dpp (no `old` operand to initialise): one v_mov_b32_dpp per half; bound_ctrl -> 0 for
  // lanes whose source is outside the row (row_shl/shr), never the case for ror / quad_perm
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// broadcast from lane J of the 16-lane row (one v_mov_b64_dpp row_newbcast)
template <int J>
__device__ __forceinline__ double bc(double x) {
  return __builtin_amdgcn_mov_dpp(x, 0x150 + J, 0xF, 0xF, true);
}
// lane that carries variable index j (0..11)
__host__ __device__ constexpr int lane_of(int j) { return 4 * (j / 3) + (j % 3); }
template <int j>
__device__ __forceinline__ double bcv(double x) { return bc<lane_of(j)>(x); }

__device__ __forceinline__ double row_sum(double x) {
  x += dpp<0x128>(x); // row_ror:8
  x += dpp<0x124>(x);
  x += dpp<0x122>(x);
  x += dpp<0x121>(x);
  return x;
}
// Row sum in single precision for quantities that only feed a threshold test (|z|^2 > eps, |psi| <= tol):
// v_add_f32 takes a DPP operand, so a level is one instruction instead of two moves and an add.
__device__ __forceinline__ float row_sum_f32(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x124, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x122, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x121, 0xF, 0xF, true));
  return x;
}
__device__ __forceinline__ double quad_sum(double x) {
  x += dpp<0xB1>(x); // quad_perm [1,0,3,2]
  x += dpp<0x4E>(x); // quad_perm [2,3,0,1]
  return x;
}
template <int K>
__device__ __forceinline__ double quad_bc(double x) { return dpp<K * 85>(x); } // quad_perm [K,K,K,K]

// acc += bcast_{LANE}(src) * mul in ONE instruction (v_fmac_f64_dpp, the only f64 VALU op that
// takes a DPP operand, and only row_newbcast).  kNop: `src` may have been written by the previous
// VALU instruction (DPP read-after-VALU-write needs 2 wait states; hipcc does not see inside asm).
template <int LANE, bool kNop = false>
__device__ __forceinline__ void fmac_bc(double &acc, double src, double mul) {
  if constexpr (kNop)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
}

// v_min_f64 as is: fmin() adds two canonicalising v_max per call (NaN quieting the hardware min already does)
__device__ __forceinline__ double vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ double row_min(double x) {
  x = vmin(x, dpp<0x128>(x));
  x = vmin(x, dpp<0x124>(x));
  x = vmin(x, dpp<0x122>(x));
  x = vmin(x, dpp<0x121>(x));
  return x;
}
// lowest lane of my row for which `pred` holds (16 if none)
__device__ __forceinline__ int row_first(bool pred) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
  const unsigned bits = (unsigned)(m >> (threadIdx.x & 48)) & 0xFFFFu;
  return bits ? (__ffs(bits) - 1) : 16;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl<0, N>(f); }

// 1/x and 1/sqrt(x): hardware seed + two Newton steps (1-2 ulp)
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y + y * (1.0 - x * y);
  y = y + y * (1.0 - x * y);
  return y;
}
// one Newton step: 2e-15 relative (the seed has 24 bits, tools/ubench/rcp_accuracy.hip); for ratios that are only compared
__device__ __forceinline__ double rcp_nr1(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  return fma(y, fma(-x, y, 1.0), y);
}
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y + y * (0.5 - 0.5 * x * y * y);
  y = y + y * (0.5 - 0.5 * x * y * y);
  return y;
}

// element c (0..2) of a replicated 3-vector; 0 for the spare lane
// (a chain of single-level selects: nested ?: on a lane-varying index is lowered to exec-mask control flow,
// a dozen scalar instructions and two branches per use, instead of two v_cndmask)
__device__ __forceinline__ double sel(bool p, double a, double b) { return p ? a : b; }
__device__ __forceinline__ double pick3(const double v[3], int c) {
  double r = 0.0;
  r = sel(c == 2, v[2], r);
  r = sel(c == 1, v[1], r);
  r = sel(c == 0, v[0], r);
  return r;
}

struct CoopTab { // one leg's model block in LDS
  const double *p;
  __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};

struct CoopPtrs {
  const double *q, *pos, *quat, *linvel, *angvel, *dpos, *dquat, *dlinvel, *dangvel;
  const uint8_t *stance;
  const double *normals;
  const double *wrench; // [B][6] or NULL: externally supplied (F_B, T_B)
  const uint8_t *live;  // [B] or NULL: 0 = leave this robot alone (whole tick: no command in force), nothing is written
};

// One robot per 16-lane row.  lds_tab: 256-double model table; lds_row: this robot's private
// LDS block of kCoopLdsDoubles doubles (N* export for the refinement, row export for drops); lds_nrm: the wavefront's
// table of constraint normals, kCoopNrmDoubles doubles ([row kind][lane]).
constexpr int kCoopLdsDoubles = 12 * 12 + 12;
constexpr int kCoopNrmDoubles = 11 * 64; // 5 row kinds + parked Jacobian row (3) and gravity torque (3)

template <bool kPerLeg, int kBlock = 64>
__device__ __forceinline__ void coop_robot(const DeviceParams &P, const CoopPtrs &s, int64_t irobot, bool robot_live_in,
                                           double *lds_tab, double *lds_row, double *lds_nrm,
                                           double *__restrict__ tau_out,
                                           double *__restrict__ grf_out, int32_t *__restrict__ status_out) {
  bool robot_live = robot_live_in;
  const int lr = threadIdx.x & 15;   // lane in row
  const int leg = lr >> 2, c = lr & 3;
  const bool comp = c < 3;           // carries a variable / matrix row
  const int myidx = 3 * leg + c;     // valid when comp
  const double eps = 2.220446049250313e-16;
  const double inf = INFINITY;

  QL_STAMP(0);
  // ---------------------------------------------------------------- load
  // The model table (4 x 88 doubles) goes to LDS for the leg-indexed reads below.  Its six loads per lane are
  // issued first and unconditionally (clamped index), the robot's own state right behind them, and only then
  // are the table values stored and the barrier taken: one memory round trip instead of seven in a row.
  constexpr int kTabLoads = (4 * kTabPerLeg + kBlock - 1) / kBlock;
  double tabv[kTabLoads];
#pragma unroll
  for (int j = 0; j < kTabLoads; j++) {
    const int idx = (int)threadIdx.x + kBlock * j;
    tabv[j] = P.legtab[idx < 4 * kTabPerLeg ? idx : 4 * kTabPerLeg - 1];
  }
  const int64_t i = irobot;
  double quat[4], dquat[4], pos[3], linvel[3], angvel[3], dpos[3], dlinvel[3], dangvel[3];
  {
    const double2 *a2 = reinterpret_cast<const double2 *>(s.quat + 4 * i);
    const double2 *b2 = reinterpret_cast<const double2 *>(s.dquat + 4 * i);
    double2 v = a2[0]; quat[0] = v.x; quat[1] = v.y;
    v = a2[1]; quat[2] = v.x; quat[3] = v.y;
    v = b2[0]; dquat[0] = v.x; dquat[1] = v.y;
    v = b2[1]; dquat[2] = v.x; dquat[3] = v.y;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      pos[k] = s.pos[3 * i + k]; linvel[k] = s.linvel[3 * i + k]; angvel[k] = s.angvel[3 * i + k];
      dpos[k] = s.dpos[3 * i + k]; dlinvel[k] = s.dlinvel[3 * i + k]; dangvel[k] = s.dangvel[3 * i + k];
    }
  }
  const uint32_t sm = *reinterpret_cast<const uint32_t *>(s.stance + 4 * i);
  const double qj = s.q[12 * i + (comp ? myidx : 0)];
  const uint8_t alive = s.live ? s.live[i] : (uint8_t)1;
  double wr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}; // externally supplied (F_B, T_B), if any: issued with the rest
  if (s.wrench) {
#pragma unroll
    for (int k = 0; k < 6; k++) wr[k] = s.wrench[6 * i + k];
  }
  double nWl[3] = {0.0, 0.0, 1.0}; // caller-supplied surface normal of my leg (world frame), if any
  if (kPerLeg) { nWl[0] = s.normals[12 * i + 3 * leg]; nWl[1] = s.normals[12 * i + 3 * leg + 1]; nWl[2] = s.normals[12 * i + 3 * leg + 2]; }
  // (all loads above are in flight before the first of them is consumed)
  robot_live = robot_live && alive != 0;
  const unsigned stance = robot_live ? (((sm & 0xFFu) ? 1u : 0u) | ((sm & 0xFF00u) ? 2u : 0u) |
                                        ((sm & 0xFF0000u) ? 4u : 0u) | ((sm & 0xFF000000u) ? 8u : 0u))
                                     : 0u;
  const int nS = __popc(stance);
  const bool on = ((stance >> leg) & 1u) != 0; // my leg supports
#pragma unroll
  for (int j = 0; j < kTabLoads; j++) {
    const int idx = (int)threadIdx.x + kBlock * j;
    if (idx < 4 * kTabPerLeg) lds_tab[idx] = tabv[j];
  }
  __syncthreads();

  QL_STAMP(1);
  // ---------------------------------------------------------------- wrench (replicated)
  double Rm[9], gB[3], b[6];
  double wr_d[3] = {0.0, 0.0, 0.0}, wr_dw = 1.0, wr_k = 2.0; // orientation error: vector part, scalar part, 2 alpha / sin(alpha)
  bool wr_slow = false;
  quat_to_matrix(quat, Rm);
  {
    const double gW[3] = {0.0, 0.0, -P.grav};
    irot(Rm, gW, gB);
    RobotIn in;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      in.pos[k] = pos[k]; in.linvel[k] = linvel[k]; in.angvel[k] = angvel[k];
      in.dpos[k] = dpos[k]; in.dlinvel[k] = dlinvel[k]; in.dangvel[k] = dangvel[k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { in.quat[k] = quat[k]; in.dquat[k] = dquat[k]; }
    // Computed unconditionally and without a branch, so that the leg kinematics below -- an independent chain of
    // about the same length -- sits in the same basic block and the scheduler can weave the two (a lone wavefront
    // issues a dependent instruction every 8.5 cycles, an independent one every 5.5).  The orientation error
    // -log(q_d^-1 q_m) = -k d_vec with k = 2 acos(d_w) / sqrt(1 - d_w^2) is evaluated as the series of
    // 2 asin(s) / s in s^2 = 1 - d_w^2 (ten terms, < 1e-16 below s^2 = 0.01, i.e. errors up to 0.2 rad); a row outside
    // that range gets the libm form as a correction after the kinematics, and an externally supplied wrench
    // (computeForceDistribution's arguments) replaces the result there as well.
    in.has_wrench = false;
    wr_dw = 0.0;
    {
      const double *qd = in.dquat, *qm = in.quat;
      const double aw = qd[0], ax = -qd[1], ay = -qd[2], az = -qd[3];
      wr_dw = aw * qm[0] - ax * qm[1] - ay * qm[2] - az * qm[3];
      wr_d[0] = aw * qm[1] + ax * qm[0] + ay * qm[3] - az * qm[2];
      wr_d[1] = aw * qm[2] - ax * qm[3] + ay * qm[0] + az * qm[1];
      wr_d[2] = aw * qm[3] + ax * qm[2] - ay * qm[1] + az * qm[0];
      const double s2 = 1.0 - wr_dw * wr_dw;
      double p = 34459425.0 / 3530096640.0;
      p = fma(p, s2, 2027025.0 / 175472640.0);
      p = fma(p, s2, 135135.0 / 9676800.0);
      p = fma(p, s2, 10395.0 / 599040.0);
      p = fma(p, s2, 945.0 / 42240.0);
      p = fma(p, s2, 105.0 / 3456.0);
      p = fma(p, s2, 15.0 / 336.0);
      p = fma(p, s2, 3.0 / 40.0);
      p = fma(p, s2, 1.0 / 6.0);
      p = fma(p, s2, 1.0);
      wr_k = 2.0 * p;
      wr_slow = !(s2 < 0.01) || !(wr_dw > 0.0);
    }
    {
      double e_p[3], e_v[3], e_w[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        e_p[k] = in.dpos[k] - in.pos[k];
        e_v[k] = in.dlinvel[k] - in.linvel[k];
        e_w[k] = in.dangvel[k] - in.angvel[k];
      }
      // VirtualModelController.cpp:208-231 (vertical P and D enter twice), as virtual_wrench()
      const double ff[3] = {in.dlinvel[0], in.dlinvel[1], 0.0};
      const double gfb[3] = {0.0, 0.0, P.kp_t[2] * e_p[2]};
      const double gdb[3] = {0.0, 0.0, P.kd_t[2] * e_v[2]};
      double Rep[3], Rev[3], Rff[3], fbp[3], fbd[3];
      irot(Rm, e_p, Rep); irot(Rm, e_v, Rev); irot(Rm, ff, Rff); irot(Rm, gfb, fbp); irot(Rm, gdb, fbd);
#pragma unroll
      for (int k = 0; k < 3; k++)
        b[k] = P.kp_t[k] * Rep[k] + P.kd_t[k] * Rev[k] + P.kff_t[k] * Rff[k] - P.Fg_scale * gB[k] + fbp[k] + fbd[k];
      // VirtualModelController.cpp:244-259
      const double kdw[3] = {P.kd_r[0] * e_w[0], P.kd_r[1] * e_w[1], P.kd_r[2] * e_w[2]};
      const double kfw[3] = {0.0, 0.0, P.kff_r[2] * in.dangvel[2]};
      double Rd[3], Rf[3], Tg[3];
      irot(Rm, kdw, Rd); irot(Rm, kfw, Rf);
      cross3(P.Tg_arm, gB, Tg);
#pragma unroll
      for (int k = 0; k < 3; k++) b[3 + k] = P.kp_r[k] * (-wr_k * wr_d[k]) + Rd[k] + Rf[k] - Tg[k];
    }
  }

  QL_STAMP(2);
  // ---------------------------------------------------------------- leg kinematics, 4 lanes per leg
  // lane c holds row c of the cumulative rotation and component c of every position
  const CoopTab tab{lds_tab + kTabPerLeg * leg};
  double sj, cj;
  sincos_reduced(qj, sj, cj);
  double Rc[3] = {c == 0 ? 1.0 : 0.0, c == 1 ? 1.0 : 0.0, c == 2 ? 1.0 : 0.0};
  double pc = 0.0;
  double zax[3], pj[3], Hk[4]; // component c of joint axes, joint origins, m_k * com_k
#pragma unroll
  for (int k = 0; k < 4; k++) {
    double R0[9], t[3], mcom[3];
#pragma unroll
    for (int a = 0; a < 9; a++) R0[a] = tab[kTabR0 + 9 * k + a];
#pragma unroll
    for (int a = 0; a < 3; a++) { t[a] = tab[kTabXyz + 3 * k + a]; mcom[a] = tab[kTabMcom + 3 * k + a]; }
    const double m = tab[kTabMass + k];
    double Rs[9];
    if (k < 3) {
      const double sk = k == 0 ? quad_bc<0>(sj) : k == 1 ? quad_bc<1>(sj) : quad_bc<2>(sj);
      const double ck = k == 0 ? quad_bc<0>(cj) : k == 1 ? quad_bc<1>(cj) : quad_bc<2>(cj);
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
    pc += Rc[0] * t[0] + Rc[1] * t[1] + Rc[2] * t[2];
    double Rn[3];
#pragma unroll
    for (int a = 0; a < 3; a++) Rn[a] = Rc[0] * Rs[a] + Rc[1] * Rs[3 + a] + Rc[2] * Rs[6 + a];
    Rc[0] = Rn[0]; Rc[1] = Rn[1]; Rc[2] = Rn[2];
    Hk[k] = m * pc + (Rn[0] * mcom[0] + Rn[1] * mcom[1] + Rn[2] * mcom[2]);
    if (k < 3) { zax[k] = Rn[2]; pj[k] = pc; }
  }
  const double foot = (comp && on) ? pc : 0.0; // component c of my leg's foot position
  // cross product component c of (a x b) for quad-distributed a, b: a_{c+1} b_{c+2} - a_{c+2} b_{c+1}
  const auto qcross = [&](double a, double bq) -> double {
    const double a1 = dpp<0xC9>(a), a2 = dpp<0xD2>(a);   // quad_perm [1,2,0,3], [2,0,1,3]
    const double b1 = dpp<0xC9>(bq), b2 = dpp<0xD2>(bq);
    return a1 * b2 - a2 * b1;
  };
  double Jrow[3], Gq[3]; // Jrow[i] = J[c][i];  Gq[i] replicated in the quad
  {
    const double my_g = pick3(gB, c);
    double Hs = Hk[3];
    double M = tab[kTabMass + 3];
#pragma unroll
    for (int k = 2; k >= 0; k--) {
      Hs += Hk[k];
      M += tab[kTabMass + k];
      const double d = pc - pj[k];
      const double jc = qcross(zax[k], d); // unconditionally: a DPP op under a lane condition becomes a branch
      Jrow[k] = sel(comp, jc, 0.0);
      const double h = Hs - M * pj[k];
      const double zh = qcross(zax[k], h);
      Gq[k] = -quad_sum(sel(comp, my_g * zh, 0.0));
    }
  }

  if (__builtin_amdgcn_ballot_w64(wr_slow) != 0ull) { // a large orientation error: the libm form of the factor
    const double s2 = 1.0 - wr_dw * wr_dw;
    double k = 2.0;
    if (s2 >= 1e-12) k = 2.0 * acos(wr_dw) / sqrt(s2);
    const double dk = wr_slow ? k - wr_k : 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++) b[3 + a] -= P.kp_r[a] * dk * wr_d[a];
  }
  if (s.wrench) {
#pragma unroll
    for (int k = 0; k < 6; k++) b[k] = wr[k];
  }
  // Jacobian row and gravity torque are not needed before the torques at the very end: parked in LDS ([k][lane]) so
  // that the kernel stays within 256 registers without a spill to scratch memory
#pragma unroll
  for (int k = 0; k < 3; k++) {
    lds_nrm[64 * (5 + k) + ((int)threadIdx.x & 63)] = Jrow[k];
    lds_nrm[64 * (8 + k) + ((int)threadIdx.x & 63)] = Gq[k];
  }

  QL_STAMP(3);
  // ---------------------------------------------------------------- friction pyramid of my leg
  double myn = 0.0, myt1 = 0.0, myt2 = 0.0; // component c of n, t1, t2 (base frame)
  double nb[3], t1[3], t2[3];               // the whole vectors of my leg
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
    myn = pick3(nb, c); myt1 = pick3(t1, c); myt2 = pick3(t2, c);
  }
  const double mu = P.mu, f_min = P.f_min;

  QL_STAMP(4);
  // ---------------------------------------------------------------- G row, g0, H = G^-1 (Gauss-Jordan)
  double Gm[12], H[12], g0;
  double c1 = 0.0, c2 = 0.0;
  {
    // foot positions of all legs, replicated
    double r[4][3];
    static_for<12>([&](auto J) { constexpr int j = J; r[j / 3][j % 3] = bcv<j>(foot); });
    const double rl[3] = {quad_bc<0>(foot), quad_bc<1>(foot), quad_bc<2>(foot)};
    // a = r_leg x e_c  (column c of skew(r_leg))
    const double a[3] = {sel(c == 1, -rl[2], sel(c == 2, rl[1], 0.0)), sel(c == 0, rl[2], sel(c == 2, -rl[0], 0.0)),
                         sel(c == 0, -rl[1], sel(c == 1, rl[0], 0.0))};
    const double sa[3] = {P.S[3] * a[0], P.S[4] * a[1], P.S[5] * a[2]};
    const double Sfc = pick3(P.S, c);
    const bool row_on = comp && on;
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
    // rows of legs that do not support are padding: unit diagonal (they never meet a stance row)
#pragma unroll
    for (int j = 0; j < 12; j++)
      if (comp && j == myidx) Gm[j] += row_on ? P.w_reg : 1.0;
    const double Fc = pick3(b, c);
    const double ST[3] = {P.S[3] * b[3], P.S[4] * b[4], P.S[5] * b[5]};
    const double g0v = -(Sfc * Fc + (a[0] * ST[0] + a[1] * ST[1] + a[2] * ST[2]));
    g0 = sel(row_on, g0v, 0.0);
#pragma unroll
    for (int j = 0; j < 12; j++) H[j] = Gm[j];
    // trace(G) over the stance block
    {
      double diag = 0.0;
#pragma unroll
      for (int j = 0; j < 12; j++) diag = (j == myidx) ? Gm[j] : diag;
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
    static_for<12>([&](auto K) {
      constexpr int k = K;
      bad = bad || !(d > 0.0);
      const double p = rcp_nr1(d);
      const bool piv = comp && (myidx == k);
      my_pivot = piv ? d : my_pivot;
      const double f = piv ? (1.0 - p) : H[k] * p;
      const double nf = -f;
      if constexpr (k < 11) {
        fmac_bc<lane_of(k), true>(H[k + 1], H[k + 1], nf);
        d = bcv<k + 1>(H[k + 1]);
      }
      static_for<12>([&](auto J) {
        constexpr int j = J;
        if constexpr (j != k && j != k + 1) fmac_bc<lane_of(k), (k == 11 && j == 0)>(H[j], H[j], nf);
      });
      H[k] = piv ? p : nf;
    });
    // c2 = trace(J) = sum over the stance rows of 1/sqrt(pivot): one rsqrt per lane instead of one per pivot
    // (it only feeds the termination tolerance psi_tol)
    const double rp = rsqrt_nr(my_pivot);
    c2 = row_sum(sel(row_on, rp, 0.0));
    if (bad && nS > 0) {
      if (lr == 0 && robot_live) status_out[i] = kStatusNotPd;
      if (comp && robot_live && !P.keep_on_failure) { tau_out[12 * i + myidx] = 0.0; if (grf_out) grf_out[12 * i + myidx] = 0.0; }
      return;
    }
  }

  QL_STAMP(5);
  // ---------------------------------------------------------------- x0 = -H g0
  double x = 0.0;
  {
    const double ng0 = -g0;
    double xa[3] = {0.0, 0.0, 0.0};
    static_for<12>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(xa[j % 3], ng0, H[j]); });
    x = (xa[0] + xa[1]) + xa[2];
  }

  QL_STAMP(6);
#ifdef QLAMD_COOP_V1
  // ---------------------------------------------------------------- active-set loop
  // Slots are NOT compacted on a drop: a freed slot lane is reused by the next add (the order of the
  // slots only breaks exact ties in the blocking-constraint search).
  double Ns[12];
#pragma unroll
  for (int j = 0; j < 12; j++) Ns[j] = 0.0;
  double u = 0.0;            // multiplier of slot lr
  int idk = 0;               // constraint id of slot lr
  unsigned used = 0;         // bit k set <=> slot lane k holds an active constraint
  int q = 0, iters = 0, status = kStatusOk;
  unsigned act_mask = 0, excl = 0;
  const double psi_tol = (double)(5 * nS) * eps * c1 * c2 * 100.0;
  double rnorm2 = 1.0; // R_norm^2
  bool done = (nS == 0), need_select = true, fresh = true;
  int ip = 0, pleg = 0, pt = 0;
  double sp = 0.0, ucand = 0.0;
  // per-lane constraint coefficients: lane c of a quad evaluates friction row t = c+1
  const double fa = c == 0 ? 1.0 : c == 1 ? -1.0 : 0.0, fb = c == 2 ? 1.0 : c == 3 ? -1.0 : 0.0;

  // slacks at x: s_min (replicated in the quad) and this lane's friction row
  const auto slacks = [&](double xx, double &s_min, double &s_fric) {
    const double dn = quad_sum(myn * xx), d1 = quad_sum(myt1 * xx), d2 = quad_sum(myt2 * xx);
    s_min = dn - f_min;
    s_fric = mu * dn + fa * d1 + fb * d2;
  };
  // component c of the normal of constraint type t on my leg (0 on the spare lane)
  // the five normals of my leg, component c: a table + selects keeps this branch-free
  const double nrm0 = myn, nrm1 = mu * myn + myt1, nrm2 = mu * myn - myt1, nrm3 = mu * myn + myt2, nrm4 = mu * myn - myt2;
  const auto my_normal = [&](int t) -> double {
    const double a = (t & 1) ? nrm1 : nrm2, b = (t & 1) ? nrm3 : nrm4;
    const double f = (t <= 2) ? a : b;
    return t == 0 ? nrm0 : f;
  };

  for (int tick = 0; tick < 40 * kMaxOuter; tick++) {
    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
    if (!done && need_select) {
      if (fresh) { iters++; excl = 0; }
      double s_min, s_fric;
      slacks(x, s_min, s_fric);
      const double viol = vmin(0.0, s_fric) + sel(c == 0, vmin(0.0, s_min), 0.0);
      const double psi = (double)row_sum_f32((float)sel(on, viol, 0.0)); // only compared with psi_tol below
      const unsigned blocked = act_mask | excl;
      // candidates of this lane: friction row 5*leg + c + 1 and (lane c == 0) the minimum-force row
      // 5*leg.  Exact ties go to the lowest lane (the reference takes the lowest row index; either
      // way the minimiser is the same, only the path differs).
      const int idf = 5 * leg + c + 1, idm = 5 * leg;
      double v = sel(on && !((blocked >> idf) & 1u) && s_fric < 0.0, s_fric, inf);
      const bool cand_min = on && c == 0 && !((blocked >> idm) & 1u) && s_min < 0.0 && s_min <= v;
      v = sel(cand_min, s_min, v);
      const double vbest = row_min(v);
      const int wl = row_first(v == vbest && v < 0.0);
      const bool wmin = ((unsigned)(__builtin_amdgcn_ballot_w64(cand_min) >> ((threadIdx.x & 48) + (wl & 15))) & 1u) != 0;
      v = vbest;
      const int key = 5 * (wl >> 2) + (wmin ? 0 : (wl & 3) + 1);
      const bool feasible = fresh && (fabs(psi) <= psi_tol); // QuadProg++.cc:246-250
      const bool stop = feasible || !(v < 0.0) || iters > kMaxOuter; // :271-274
      status = (stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      done = stop;
      ip = stop ? ip : key;
      pleg = id_leg(ip); pt = ip - 5 * pleg;
      sp = sel(stop, sp, v);
      ucand = sel(stop, ucand, 0.0);
      need_select = stop;
    }
    if (!done) {
      // ---- directions: z = H n_p (lane i), r = N* n_p (slot lane k)
      const double npj = (leg == pleg) ? my_normal(pt) : 0.0;
      // three partial sums per product: consecutive dependent FMAs are 6 instructions apart
      double za[3] = {0.0, 0.0, 0.0}, ra[3] = {0.0, 0.0, 0.0};
      static_for<12>([&](auto J) {
        constexpr int j = J;
        fmac_bc<lane_of(j), j == 0>(za[j % 3], npj, H[j]);
        fmac_bc<lane_of(j)>(ra[j % 3], npj, Ns[j]);
      });
      const double z = (za[0] + za[1]) + za[2], r = (ra[0] + ra[1]) + ra[2];
      const bool slot = (used & lanebit) != 0u;
      const double zn = row_sum(z * npj);
      const float zf = (float)z;
      const double zz = (double)row_sum_f32(zf * zf); // only compared with eps below
      // ---- step lengths, QuadProg++.cc:304-331
      const double ur = u * rcp_nr(r);
      const double ratio = sel(slot && r > 0.0, ur, inf);
      const double t1 = row_min(ratio);
      const int lpos = row_first(ratio == t1 && ratio < inf);
      const double t2v = -sp * rcp_nr(zn);
      const bool exhausted = q >= 3 * nS; // empty null space: z is exactly 0 in the reference
      const double t2 = sel((int)(!exhausted) & (int)(fabs(zz) > eps) & (int)(!(t2v < 0.0)), t2v, inf);
      const double t = vmin(t1, t2);
      // what happens this tick (all row-uniform)
      const bool infeasible = !(t < inf);                          // :339-344
      const bool dual_only = (t2 >= inf);
      const bool full = !infeasible && !dual_only && (t2 <= t1);   // :384
      // add_constraint fails when |R_qq| = sqrt(z'n_p) <= eps * R_norm (:392); compared squared
      const bool degenerate = full && !(zn > eps * eps * rnorm2);
      const bool is_add = full && !degenerate;
      const bool is_drop = !infeasible && !full;                   // partial or dual-only step
      if (infeasible) { status = kStatusInfeasible; done = true; }
      // ---- the step
      const double tp = (infeasible || dual_only || degenerate) ? 0.0 : t;
      const double td = (infeasible || degenerate) ? 0.0 : t;
      x += tp * z;
      u -= sel(slot, td * r, 0.0);
      ucand += td;
      sp += tp * zn; // slack of ip after a partial step (:436-440, linear in t)
      // ---- rank-one update of H and N*:  H[j] += hc * v_j,  N*[j] += nc * v_j
      // add (predicated, no branch):  H -= z z'/d;  N* <- [N* - r z'/d ; z'/d], the new row goes to the
      // lowest free slot lane.  A numerically dependent normal is skipped and selection repeated.
      const int newlane = __ffs(~used & 0xFFFu) - 1;
      const bool newslot = is_add && (lr == newlane);
      double vec = is_add ? z * rcp_nr(zn) : 0.0;
      double hc = is_add ? -z : 0.0;
      double nc = sel(newslot, 1.0, sel(is_add && slot, -r, 0.0));
      u = newslot ? ucand : u;
      idk = newslot ? ip : idk;
      used |= is_add ? (1u << newlane) : 0u;
      act_mask |= is_add ? (1u << ip) : 0u;
      rnorm2 = is_add ? fmax(rnorm2, zn) : rnorm2;
      q += is_add ? 1 : 0;
      excl |= degenerate ? (1u << ip) : 0u;
      need_select = need_select || full;
      fresh = is_add ? true : (degenerate ? false : fresh);
      if (is_drop) {
        // n~ = row lpos of N*: through LDS so that lane (leg,c) gets element myidx of it
        if (lr == lpos) {
#pragma unroll
          for (int j = 0; j < 12; j++) lds_row[144 + j] = Ns[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
        const double nt_me = comp ? lds_row[144 + myidx] : 0.0;
        // Gn = G n~ (lane i), e = n~'G n~, coef_k = N*_k . Gn / e
        double Gn = 0.0;
        static_for<12>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(Gn, nt_me, Gm[j]); });
        const double einv = rcp_nr(row_sum(nt_me * Gn));
        double coef = 0.0;
        static_for<12>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(coef, Gn, Ns[j]); });
        // H += n~ n~'/e;  N* -= coef n~'  (row lpos becomes exactly 0: coef = 1 there)
        vec = nt_me;
        hc = nt_me * einv;
        nc = -coef * einv;
        const int drop_id = __shfl(idk, lpos, 16);
        act_mask &= ~(1u << drop_id);
        used &= ~(1u << lpos);
        if (lr == lpos) u = 0.0;
        q--;
      }
      static_for<12>([&](auto J) {
        constexpr int j = J;
        fmac_bc<lane_of(j), j == 0>(H[j], vec, hc);
        fmac_bc<lane_of(j)>(Ns[j], vec, nc);
      });
      if (is_drop && lr == lpos) {
#pragma unroll
        for (int j = 0; j < 12; j++) Ns[j] = 0.0;
      }
    }
  }

#else
  // ---------------------------------------------------------------- active-set loop
  // One pass of the loop = one step of the dual method (an add or a drop) followed by the rank-one update of H and
  // N*, with the selection of the next violated constraint (QuadProg++.cc:252-274) computed on the new x in the
  // shadow of that update: the two are independent, so the broadcast chain of the update fills the wait states of
  // the selection's cross-lane reduction and vice versa.  Slots are NOT compacted on a drop: a freed slot lane is
  // reused by the next add (the order of the slots only breaks exact ties in the blocking-constraint search).
  //
  // A lone wavefront issues one instruction every ~4.5 cycles whatever its kind (tools/ubench/issue_model.hip), so a
  // pass costs what it has instructions.  The four robots of a wavefront take different branches of the method, which
  // makes every state update a predicated select; but the launch lasts as long as its slowest robot, which spends
  // most of its passes as the only live row of its wavefront.  So the tail of a pass exists three times: all live
  // rows add (no predication, selection follows), all live rows drop (no selection), and the general predicated form.
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
  double Ns[12];
#pragma unroll
  for (int j = 0; j < 12; j++) Ns[j] = 0.0;
  double u = 0.0;            // multiplier of slot lr (free lanes: never read)
  int idk = 0;               // constraint id of slot lr
  unsigned used = 0;         // bit k set <=> slot lane k holds an active constraint
  int q = 0, iters = 0, status = kStatusOk;
  unsigned act_mask = 0, excl = 0;
  const double psi_tol = (double)(5 * nS) * eps * c1 * c2 * 100.0;
  double rnorm2 = 1.0; // R_norm^2
  bool done = (nS == 0);
  int ip = 0;
  double sp = 0.0, ucand = 0.0, npj = 0.0;
  // rows this lane evaluates
  const double fa = c == 0 ? 1.0 : c == 1 ? -1.0 : 0.0, fb = c == 2 ? 1.0 : c == 3 ? -1.0 : 0.0;
  const double Wf0 = mu * nb[0] + (fa * t1[0] + fb * t2[0]), Wf1 = mu * nb[1] + (fa * t1[1] + fb * t2[1]),
               Wf2 = mu * nb[2] + (fa * t1[2] + fb * t2[2]);
  const unsigned maskf = on ? (1u << (5 * leg + c + 1)) : 0u, maskm = (on && c == 0) ? (1u << (5 * leg)) : 0u;
  const unsigned tagf = (unsigned)lr << 1, tagm = tagf | 1u;
  const int row_addr = ((int)threadIdx.x & 48) << 2; // ds_bpermute byte address of lane 0 of my row
  const unsigned lanebit = 1u << lr;
  // table of normals: entry [kind][lane] = component c of my leg's row of that kind (0 minimum force, 1..4 friction)
  {
    const double nrm[5] = {myn, mu * myn + myt1, mu * myn - myt1, mu * myn + myt2, mu * myn - myt2};
#pragma unroll
    for (int k = 0; k < 5; k++) lds_nrm[64 * k + ((int)threadIdx.x & 63)] = nrm[k];
  }
  const auto slacks = [&](double xx, double &s_min, double &s_fric) {
    const double x0 = quad_bc<0>(xx), x1 = quad_bc<1>(xx), x2 = quad_bc<2>(xx);
    s_fric = fma(Wf2, x2, fma(Wf1, x1, Wf0 * x0));
    s_min = fma(nb[2], x2, fma(nb[1], x1, fma(nb[0], x0, -f_min)));
  };
  const auto umax_dpp = [](unsigned k, auto Ctrl) -> unsigned {
    constexpr int ctrl = decltype(Ctrl)::value;
    const unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, ctrl, 0xF, 0xF, true);
    return k > o ? k : o;
  };

  // Update of H and N* with the vectors of the step just taken (H[j] += hc * vec_j, N*[j] += nc * vec_j) and
  // selection of the next constraint at the new x, in one block so that the scheduler can weave the two (and the
  // bookkeeping of the step in front of them) together.  kMode 0: before the first step (no update; every live row
  // selects).  kMode 1: general -- rows in `resel` select (`fresh`: after an add, :252-262), the others keep their
  // candidate.  kMode 2: every live row has just added a constraint.
  double vec = 0.0, hc = 0.0, nc = 0.0;
  const auto update_and_select = [&](auto Mode, bool resel, bool fresh) {
    constexpr int kMode = decltype(Mode)::value;
    constexpr bool kUpd = kMode != 0;
    if constexpr (kMode == 1) {
      iters += (resel && fresh) ? 1 : 0;
      excl = (resel && fresh) ? 0u : excl;
    } else {
      iters += 1;
      excl = 0u;
    }
    const unsigned avail = ~(act_mask | excl);
    double s_min, s_fric;
    slacks(x, s_min, s_fric);
    unsigned kf = __float_as_uint((float)s_fric), km = __float_as_uint((float)s_min);
    kf = ((avail & maskf) != 0u && s_fric < 0.0) ? ((kf & ~31u) | tagf) : 0u;
    km = ((avail & maskm) != 0u && s_min < 0.0) ? ((km & ~31u) | tagm) : 0u;
    const double myv = km > kf ? s_min : s_fric; // the slack behind this lane's key
    unsigned key = km > kf ? km : kf;
    if constexpr (kUpd) {
      static_for<3>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); });
    }
    key = umax_dpp(key, std::integral_constant<int, 0x128>{});
    if constexpr (kUpd) {
      static_for<3>([&](auto J) { constexpr int j = J + 3; fmac_bc<lane_of(j)>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); });
    }
    key = umax_dpp(key, std::integral_constant<int, 0x124>{});
    if constexpr (kUpd) {
      static_for<2>([&](auto J) { constexpr int j = J + 6; fmac_bc<lane_of(j)>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); });
    }
    key = umax_dpp(key, std::integral_constant<int, 0x122>{});
    if constexpr (kUpd) {
      static_for<2>([&](auto J) { constexpr int j = J + 8; fmac_bc<lane_of(j)>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); });
    }
    key = umax_dpp(key, std::integral_constant<int, 0x121>{});
    // the chosen row: lane and kind from the low bits, its slack from its lane, its normal from the table
    const int wl = (int)(key >> 1) & 15;
    const int addr = row_addr + (wl << 2);
    const int vlo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(myv));
    const int vhi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(myv));
    const int key_kind = (key & 1u) ? 0 : (wl & 3) + 1;
    const int key_ip = 5 * (wl >> 2) + key_kind;
    const double np_tab = lds_nrm[64 * key_kind + ((int)threadIdx.x & 63)];
    if constexpr (kUpd) {
      static_for<2>([&](auto J) { constexpr int j = J + 10; fmac_bc<lane_of(j)>(H[j], vec, hc); fmac_bc<lane_of(j)>(Ns[j], vec, nc); });
    }
    const double np_new = sel((wl >> 2) == leg, np_tab, 0.0);
    const bool any = (int)key < 0;                       // a violated row that may enter
    const double v = __hiloint2double(vhi, vlo);
    // feasibility, QuadProg++.cc:246-250: only when the worst slack is within the tolerance can the sum be
    bool feasible = false;
    const bool close = (kMode != 1 || (resel && fresh)) && any && !(v < -psi_tol);
    if (__builtin_amdgcn_ballot_w64(close) != 0ull) {
      const double viol = vmin(0.0, s_fric) + sel(c == 0, vmin(0.0, s_min), 0.0);
      const double psi = (double)row_sum_f32((float)sel(on, viol, 0.0));
      feasible = close && (fabs(psi) <= psi_tol);
    }
    const bool stop = !any || feasible || iters > kMaxOuter; // :271-274
    if constexpr (kMode == 1) {
      status = (resel && stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      done = done || (resel && stop);
      const bool take = resel && !stop;
      ip = take ? key_ip : ip;
      sp = sel(take, v, sp);
      ucand = sel(take, 0.0, ucand);
      npj = sel(take, np_new, npj);
    } else { // every row here selects: what a stopping row is left with is never read
      status = (stop && iters > kMaxOuter) ? kStatusMaxIter : status;
      done = done || stop;
      ip = key_ip; sp = v; ucand = 0.0; npj = np_new;
    }
  };
  const auto update_only = [&]() {
    static_for<12>([&](auto J) {
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
      for (int j = 0; j < 12; j++) lds_row[144 + j] = Ns[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
    const double nt_me = comp ? lds_row[144 + myidx] : 0.0;
    const int drop_id = __shfl(idk, lpos, 16);
    double Gn = 0.0;
    static_for<12>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(Gn, nt_me, Gm[j]); });
    const double einv = rcp_nr1(row_sum(nt_me * Gn));
    drop_einv = einv;
    double coef = 0.0;
    static_for<12>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(coef, Gn, Ns[j]); });
    vec = nt_me;
    hc = nt_me * einv;
    nc = -coef * einv;
    return drop_id;
  };

  update_and_select(std::integral_constant<int, 0>{}, true, true);

  // Loop structure.  A pass = directions and step lengths, then what the step is.  Passes in which EVERY live row adds
  // its constraint run in an inner loop that is one straight path: unpredicated bookkeeping, then update + selection.
  // It is left as soon as some row drops, fails or is infeasible; that pass is finished by the tail below (all rows
  // drop: unpredicated, no selection; otherwise the general predicated form) and the inner loop is entered again.
  // (Alternative tails inside ONE loop cost 24 register copies of H and N* on its back edge; and the launch lasts as
  // long as its slowest robot, which is alone in its wavefront for most of its passes.)
  // Terminates: at most kMaxOuter adds, every drop undoes an earlier add, a failed add bans its row until the next add.
  double z = 0.0, r = 0.0, zn = 0.0, zinv = 0.0, t = 0.0, tl1 = 0.0, tl2 = 0.0, ratio = 0.0;
  bool have_dirs = false; // wave-uniform: the directions of the pass at hand follow from the drop just made
  for (;;) {
    bool is_add = false;
    while (!done) {
      if (!have_dirs) {
        // ---- directions: z = H n_p (lane i), r = N* n_p (slot lane k; 0 on free lanes, whose rows are 0)
        // three partial sums per product: consecutive dependent FMAs are 6 instructions apart
        double za[3] = {0.0, 0.0, 0.0}, ra[3] = {0.0, 0.0, 0.0};
        static_for<12>([&](auto J) {
          constexpr int j = J;
          fmac_bc<lane_of(j), j == 0>(za[j % 3], npj, H[j]);
          fmac_bc<lane_of(j)>(ra[j % 3], npj, Ns[j]);
        });
        z = (za[0] + za[1]) + za[2];
        r = (ra[0] + ra[1]) + ra[2];
        zn = row_sum(z * npj);
      }
      have_dirs = false;
      const bool slot = (used & lanebit) != 0u;
      const float zf = (float)z;
      const double zz = (double)row_sum_f32(zf * zf); // only compared with eps below
      // ---- step lengths, QuadProg++.cc:304-331
      const double ur = u * rcp_nr1(r);
      ratio = sel(slot && r > 0.0, ur, inf);
      tl1 = row_min(ratio);
      zinv = rcp_nr(zn);
      const double t2v = -sp * zinv;
      const bool exhausted = q >= 3 * nS; // empty null space: z is exactly 0 in the reference
      tl2 = sel((int)(!exhausted) & (int)(fabs(zz) > eps) & (int)(!(t2v < 0.0)), t2v, inf);
      t = vmin(tl1, tl2);
      // a full step (:384) whose constraint can be added (:392: |R_qq| = sqrt(z'n_p) > eps * R_norm, compared squared)
      is_add = (tl2 < inf) && (tl2 <= tl1) && (zn > eps * eps * rnorm2);
      if (__builtin_amdgcn_ballot_w64(!is_add) != 0ull) break;
      // ---- every live row takes a full step and adds its constraint: H -= z z'/d, N* <- [N* - r z'/d ; z'/d],
      // the new row goes to the lowest free slot lane
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
      act_mask |= 1u << ip;
      rnorm2 = vmax(rnorm2, zn);
      q += 1;
      update_and_select(std::integral_constant<int, 2>{}, true, true);
    }
    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
    if (__builtin_amdgcn_ballot_w64(!done && !(tl1 < tl2)) == 0ull) {
      // ---- every live row drops a constraint (t1 < t2: partial step, or dual step only when t2 is infinite)
      if (!done) {
        const double tp = (tl2 >= inf) ? 0.0 : t;
        x += tp * z;
        u = fma(-t, r, u);
        ucand += t;
        sp += tp * zn; // slack of ip after a partial step (:436-440, linear in t)
        const int lpos = row_first(ratio == tl1 && ratio < inf);
        const double r_lpos = __shfl(r, lpos, 16);
        const int drop_id = drop_vectors(lpos);
        act_mask &= ~(1u << drop_id);
        used &= ~(1u << lpos);
        q--;
        update_only();
        if (lr == lpos) {
#pragma unroll
          for (int j = 0; j < 12; j++) Ns[j] = 0.0;
        }
        // the same candidate continues with the working set one smaller: with H' = H + n~ n~'/e and N*' = N* - c n~'/e
        // its directions are z' = z + (n~/e) r_k, r' = r - (c/e) r_k, z'n = zn + r_k^2/e (r_k = n~'n_p = r of the
        // dropped slot) -- no need for the 24 broadcasts of the next pass
        z = fma(hc, r_lpos, z);
        r = fma(nc, r_lpos, r);
        zn = fma(r_lpos * r_lpos, drop_einv, zn);
      }
      have_dirs = true;
    } else if (!done) {
      // ---- the pass of the live rows in general form (all row-uniform)
      const bool infeasible = !(t < inf);                          // :339-344
      const bool dual_only = (tl2 >= inf);
      const bool full = !infeasible && !dual_only && (tl2 <= tl1);   // :384
      const bool degenerate = full && !is_add;
      const bool is_drop = !infeasible && !full;                   // partial or dual-only step
      if (infeasible) { status = kStatusInfeasible; done = true; }
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
      if (is_drop && lr == lpos) {
#pragma unroll
        for (int j = 0; j < 12; j++) Ns[j] = 0.0;
      }
    }
  }
#endif

  QL_STAMP(7);
  // ---------------------------------------------------------------- refinement on the final working set
  if (!done) status = kStatusMaxIter;
  if (status == kStatusOk && q > 0) {
    // export N* through LDS once: lane (leg,c) needs column myidx of N*
    if (lr < 12) {
#pragma unroll
      for (int j = 0; j < 12; j++) lds_row[12 * lr + j] = Ns[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double NsT[12]; // N*[k][myidx], k = 0..11
#pragma unroll
    for (int k = 0; k < 12; k++) NsT[k] = comp ? lds_row[12 * k + myidx] : 0.0;
    const int lg = id_leg(idk), tt = idk - 5 * lg;
    const bool myslot = (used >> lr) & 1u;
    const int src = myslot ? (4 * lg + (tt == 0 ? 0 : tt - 1)) : 0;
    for (int pass = 0; pass < P.refine_passes; pass++) {
      // (1) reduced gradient: x -= H (G x + g0)
      double grad = g0;
      static_for<12>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(grad, x, Gm[j]); });
      double corr = 0.0;
      static_for<12>([&](auto J) { constexpr int j = J; fmac_bc<lane_of(j), j == 0>(corr, grad, H[j]); });
      x -= corr;
      // (2) constraint residuals rho_k = b_k - n_k'x on slot lanes; x += N*' rho
      double s_min, s_fric;
      slacks(x, s_min, s_fric);
      const double vm = __shfl(s_min, src, 16), vf = __shfl(s_fric, src, 16);
      const double rho = sel(myslot, -sel(tt == 0, vm, vf), 0.0);
      double dx = 0.0;
      static_for<12>([&](auto K) { constexpr int k = K; fmac_bc<k, k == 0>(dx, rho, NsT[k]); });
      x += dx;
    }
  }

  QL_STAMP(8);
  // ---------------------------------------------------------------- torques (phase C)
  {
    const bool live = on && status == kStatusOk;
    const double fx = live ? -x : 0.0;
    double Jr[3], Gr[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      Jr[k] = lds_nrm[64 * (5 + k) + ((int)threadIdx.x & 63)];
      Gr[k] = lds_nrm[64 * (8 + k) + ((int)threadIdx.x & 63)];
    }
    const double t0 = quad_sum(sel(comp, Jr[0] * fx, 0.0)) + Gr[0];
    const double t1 = quad_sum(sel(comp, Jr[1] * fx, 0.0)) + Gr[1];
    const double t2 = quad_sum(sel(comp, Jr[2] * fx, 0.0)) + Gr[2];
    double t = sel(c == 0, t0, sel(c == 1, t1, t2));
    t = t > P.tau_max ? P.tau_max : t;
    t = t < -P.tau_max ? -P.tau_max : t;
    if (comp && robot_live && !(P.keep_on_failure && status != kStatusOk)) {
      tau_out[12 * i + myidx] = live ? t : 0.0;
      if (grf_out) grf_out[12 * i + myidx] = live ? x : 0.0;
    }
    if (lr == 0 && robot_live) status_out[i] = status;
  }
  QL_STAMP(9);
  QL_STAMP(10);
}

} // namespace coop
} // namespace qlamd
