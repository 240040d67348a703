// Cross-lane building blocks of the lane-cooperative kernels: DPP moves inside a 16-lane row, row / quad reductions,
// the broadcast-FMA, reciprocal helpers.  Device-only (gfx950).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "balance_core.hpp"

namespace qlamd {
namespace coop {

// Diagnostic build only (-DQLAMD_STAMPS): s_memtime at segment boundaries of wave 0, read back through
// qlamd_debug_stamps (balance unit) / qlamd_debug_stamps_pose / _tick / _wholebody: every translation unit keeps a stamp
// buffer of its own, so that the diagnostic build compiles the units exactly as the shipped library does (same flags per
// unit, build.py).  Never compiled into the shipped library.
#ifdef QLAMD_STAMPS
static __device__ unsigned long long g_stamps[64];
#define QLAMD_STAMPS_ACCESSOR(name)                                                                                      \
  extern "C" int name(unsigned long long *out, int n) {                                                                  \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(::qlamd::coop::g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1; \
  }
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
#define QLAMD_STAMPS_ACCESSOR(name)
#define QL_STAMP(k)
#define QL_SEG_DECL
#define QL_SEG_START
#define QL_SEG(k)
#define QL_SEG_STORE
#endif
// -DQLAMD_BLOCK_STAMPS alone: the workgroup stamps without the segment stamps above, whose waits and scheduling barriers
// make the solver 2.7 x slower -- a launch's phases at (nearly) the shipped kernel's own pace
#if defined(QLAMD_STAMPS) || defined(QLAMD_BLOCK_STAMPS)
// one stamp per WORKGROUP (slot = which event, up to eight per unit; workgroups beyond kBlockStamps are not recorded):
// s_memrealtime, the 100 MHz counter that is one clock for the whole device (s_memtime counts per XCD), read back through
// qlamd_debug_block_stamps_*.  Which workgroup a launch waits for, and what it was doing.
constexpr int kBlockStamps = 2048;
static __device__ unsigned long long g_block_stamps[8][kBlockStamps];
#define QLAMD_BLOCK_STAMPS_ACCESSOR(name)                                                                                \
  extern "C" int name(unsigned long long *out, int slot, int n) {                                                        \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(::qlamd::coop::g_block_stamps), sizeof(unsigned long long) * n,           \
                               sizeof(unsigned long long) * ::qlamd::coop::kBlockStamps * slot) == hipSuccess ? 0 : -1; \
  }
#define QL_BLOCK_STAMP(slot)                                                                \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    if (threadIdx.x == 0 && blockIdx.x < ::qlamd::coop::kBlockStamps) ::qlamd::coop::g_block_stamps[slot][blockIdx.x] = t_; \
  } while (0)
#else
#define QLAMD_BLOCK_STAMPS_ACCESSOR(name)
#define QL_BLOCK_STAMP(slot)
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


} // namespace coop
} // namespace qlamd
