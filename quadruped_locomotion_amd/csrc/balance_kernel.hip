// HIP kernels (gfx950) for the batched balance-controller step and the C-ABI
// of include/qlamd.h.  See DESIGN.md for the data layout and launch geometry.
#include <hip/hip_runtime.h>

#include <new>
#include <stdio.h>
#include <string.h>

#include "balance_coop.hpp"
#include "pose_coop.hpp"
#include "qp_coop.hpp"
#include "wholebody_coop.hpp"
#include "balance_core.hpp"
#include "params_build.hpp"
#include "pose_core.hpp"
#include "swing_core.hpp"
#include "leg_state_core.hpp"
#include "wire_core.hpp"
#include "qlamd.h"

using namespace qlamd;

namespace {

// Per-robot run-time indexed arrays in LDS, [element][robot-in-wave]: a lane's
// bank depends on the lane only, so divergent element indices never conflict.
struct LdsScratch {
  double *base;
  int stride;
  __device__ __forceinline__ double &at(int e) { return base[e * stride]; }
};

struct StatePtrs {
  const double *q, *pos, *quat, *linvel, *angvel, *dpos, *dquat, *dlinvel, *dangvel;
  const uint8_t *stance;
  const double *normals;
  const double *wrench; // [B][6] or NULL
};

__device__ __forceinline__ void load_robot(const StatePtrs &s, int64_t i, RobotIn &in) {
  // 96-byte (q), 32-byte (quat) and 24-byte records: each lane reads its own
  // contiguous record; neighbouring lanes share cache lines, every fetched byte is used.
  const double2 *q2 = reinterpret_cast<const double2 *>(s.q + 12 * i);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double2 v = q2[k];
    in.q[2 * k] = v.x;
    in.q[2 * k + 1] = v.y;
  }
  const double2 *a2 = reinterpret_cast<const double2 *>(s.quat + 4 * i);
  const double2 *b2 = reinterpret_cast<const double2 *>(s.dquat + 4 * i);
  double2 v = a2[0]; in.quat[0] = v.x; in.quat[1] = v.y;
  v = a2[1]; in.quat[2] = v.x; in.quat[3] = v.y;
  v = b2[0]; in.dquat[0] = v.x; in.dquat[1] = v.y;
  v = b2[1]; in.dquat[2] = v.x; in.dquat[3] = v.y;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    in.pos[k] = s.pos[3 * i + k];
    in.linvel[k] = s.linvel[3 * i + k];
    in.angvel[k] = s.angvel[3 * i + k];
    in.dpos[k] = s.dpos[3 * i + k];
    in.dlinvel[k] = s.dlinvel[3 * i + k];
    in.dangvel[k] = s.dangvel[3 * i + k];
  }
  const uint32_t m = *reinterpret_cast<const uint32_t *>(s.stance + 4 * i);
  in.stance = ((m & 0xFFu) ? 1u : 0u) | ((m & 0xFF00u) ? 2u : 0u) | ((m & 0xFF0000u) ? 4u : 0u) |
              ((m & 0xFF000000u) ? 8u : 0u);
  in.has_wrench = s.wrench != nullptr;
  if (in.has_wrench) {
#pragma unroll
    for (int k = 0; k < 6; k++) in.wrench[k] = s.wrench[6 * i + k];
  }
}

struct LdsTab { // one leg's 64-double block of the model table, staged in LDS
  const double *p;
  __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};

// One wavefront per workgroup, RPW robots per wavefront (64, 16 or 4).  A small batch is spread
// over more SIMDs by lowering RPW: at 4096 robots the step is latency-bound, not throughput-bound.
//   phase A  lanes = (robot, leg) pairs: FK + Jacobian + gravity torque       -> LDS
//   phase B  lanes = robots: wrench, QP assembly, Cholesky, active-set QP      -> LDS
//   phase C  lanes = (robot, leg) pairs: tau = J'(-f) + G(q), clamp            -> HBM
// Loads and stores of phases A and C are contiguous across lanes (24-byte records in
// (robot, leg) order); phase B reads its 26 base-state doubles as per-robot records.
template <int RPW, bool kPerLeg>
__global__ __launch_bounds__(64) void balance_step_kernel(const DeviceParams *__restrict__ Pp, const StatePtrs s,
                                                          int64_t B, double *__restrict__ tau,
                                                          double *__restrict__ grf, int32_t *__restrict__ status) {
  extern __shared__ double lds[];
  double *tab = lds;                 // 256 doubles: leg model table
  double *scratch = lds + 4 * kTabPerLeg; // [kScratchDoubles][RPW]
  const DeviceParams &P = *Pp;
  const int lane = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * RPW;

#pragma unroll
  for (int i = lane; i < 4 * kTabPerLeg; i += 64) tab[i] = P.legtab[i];
  __syncthreads();

  // ---- phase A
#pragma unroll 1
  for (int item = lane; item < 4 * RPW; item += 64) {
    const int rb = item >> 2, leg = item & 3;
    const int64_t i = base + rb;
    if (i < B) {
      const bool support = s.stance[4 * i + leg] != 0;
      const double q3[3] = {s.q[12 * i + 3 * leg], s.q[12 * i + 3 * leg + 1], s.q[12 * i + 3 * leg + 2]};
      const double2 *a2 = reinterpret_cast<const double2 *>(s.quat + 4 * i);
      const double2 v0 = a2[0], v1 = a2[1];
      const double quat[4] = {v0.x, v0.y, v1.x, v1.y};
      LdsScratch scr{scratch + rb, RPW};
      phase_a_leg(LdsTab{tab + kTabPerLeg * leg}, leg, support, q3, quat, P.grav, scr);
    }
  }
  __syncthreads();

  // ---- phase B
  {
    const int64_t i = base + lane;
    if (lane < RPW && i < B) {
      RobotIn in;
      load_robot(s, i, in);
      double nw[12];
      if (kPerLeg) {
#pragma unroll
        for (int k = 0; k < 12; k++) nw[k] = s.normals[12 * i + k];
      }
      LdsScratch scr{scratch + lane, RPW};
      const QpResult r = phase_b_robot<kPerLeg>(P, in, nw, scr);
      scr.at(kScrStatus) = (double)r.status;
      status[i] = r.status;
    }
  }
  __syncthreads();

  // ---- phase C
#pragma unroll 1
  for (int item = lane; item < 4 * RPW; item += 64) {
    const int rb = item >> 2, leg = item & 3;
    const int64_t i = base + rb;
    if (i < B) {
      LdsScratch scr{scratch + rb, RPW};
      const bool live = (s.stance[4 * i + leg] != 0) && (scr.at(kScrStatus) == 0.0);
      double t[3], f[3];
      phase_c_leg(leg, live, P.tau_max, scr, t, f);
      double *to = tau + 12 * i + 3 * leg;
      to[0] = t[0]; to[1] = t[1]; to[2] = t[2];
      if (grf) {
        double *go = grf + 12 * i + 3 * leg;
        go[0] = f[0]; go[1] = f[1]; go[2] = f[2];
      }
    }
  }
}

// Latency form: 16 lanes per robot, 4 robots per wavefront (balance_coop.hpp).
template <bool kPerLeg>
__global__ __launch_bounds__(64) void balance_coop_kernel(const DeviceParams *__restrict__ Pp, const StatePtrs s,
                                                          int64_t B, double *__restrict__ tau,
                                                          double *__restrict__ grf, int32_t *__restrict__ status) {
  __shared__ double tab[4 * kTabPerLeg];
  __shared__ double rows[4 * coop::kCoopLdsDoubles];
  const DeviceParams &P = *Pp;
  const int row = threadIdx.x >> 4;
  int64_t i = (int64_t)blockIdx.x * 4 + row;
  const bool live = i < B;
  if (!live) i = B - 1;
  const coop::CoopPtrs cp{s.q, s.pos, s.quat, s.linvel, s.angvel, s.dpos, s.dquat, s.dlinvel, s.dangvel, s.stance,
                          s.normals, s.wrench};
#ifdef QLAMD_STAMPS
#pragma unroll 1
  for (int rep = 0; rep < 2; rep++) // second pass runs with a warm instruction cache
#endif
  coop::coop_robot<kPerLeg>(P, cp, i, live, tab, rows + row * coop::kCoopLdsDoubles, tau, grf, status);
}

__global__ __launch_bounds__(64) void virtual_wrench_kernel(const DeviceParams *__restrict__ Pp, const StatePtrs s,
                                                            int64_t B, double *__restrict__ wrench) {
  const DeviceParams &P = *Pp;
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= B) return;
  RobotIn in; // only the base state enters the wrench (joint positions and stance flags are not read)
  {
    const double2 *a2 = reinterpret_cast<const double2 *>(s.quat + 4 * i);
    const double2 *b2 = reinterpret_cast<const double2 *>(s.dquat + 4 * i);
    double2 v = a2[0]; in.quat[0] = v.x; in.quat[1] = v.y;
    v = a2[1]; in.quat[2] = v.x; in.quat[3] = v.y;
    v = b2[0]; in.dquat[0] = v.x; in.dquat[1] = v.y;
    v = b2[1]; in.dquat[2] = v.x; in.dquat[3] = v.y;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      in.pos[k] = s.pos[3 * i + k]; in.linvel[k] = s.linvel[3 * i + k]; in.angvel[k] = s.angvel[3 * i + k];
      in.dpos[k] = s.dpos[3 * i + k]; in.dlinvel[k] = s.dlinvel[3 * i + k]; in.dangvel[k] = s.dangvel[3 * i + k];
    }
    in.stance = 0; in.has_wrench = false;
  }
  double Rm[9], gB[3], b[6];
  quat_to_matrix(in.quat, Rm);
  const double gW[3] = {0.0, 0.0, -P.grav};
  irot(Rm, gW, gB);
  virtual_wrench(P, in, Rm, gB, b);
#pragma unroll
  for (int k = 0; k < 6; k++) wrench[6 * i + k] = b[k];
}

// The per-leg kernels below read ~60 model constants per lane.  Straight from global memory the compiler
// interleaves those reads with the arithmetic, a memory round trip every few dozen instructions; instead the
// 4 x 88-double table is staged in LDS once per block: its six loads per lane are issued first, the lane's own
// inputs right behind them, then the table is stored and the block synchronises -- one round trip in all.
struct TabStage {
  double v[6];
  __device__ __forceinline__ void issue(const DeviceParams &P) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const int idx = (int)threadIdx.x + 64 * j;
      v[j] = P.legtab[idx < 4 * kTabPerLeg ? idx : 4 * kTabPerLeg - 1];
    }
  }
  __device__ __forceinline__ void commit(double *lds_tab) const {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const int idx = (int)threadIdx.x + 64 * j;
      if (idx < 4 * kTabPerLeg) lds_tab[idx] = v[j];
    }
    __syncthreads();
  }
};
__device__ __forceinline__ void load3(const double *p, int64_t t, double o[3]) {
  o[0] = p[3 * t]; o[1] = p[3 * t + 1]; o[2] = p[3 * t + 2];
}

__global__ __launch_bounds__(64) void leg_kinematics_kernel(const DeviceParams *__restrict__ Pp,
                                                            const double *__restrict__ q,
                                                            const double *__restrict__ quat, int64_t B,
                                                            double *__restrict__ foot, double *__restrict__ jac,
                                                            double *__restrict__ grav) {
  // one lane per (robot, leg)
  __shared__ double tab[4 * kTabPerLeg];
  const DeviceParams &P = *Pp;
  TabStage ts;
  ts.issue(P);
  const int64_t t0 = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = t0 < 4 * B;
  const int64_t t = live ? t0 : 4 * B - 1;
  const int64_t i = t >> 2;
  const int leg = (int)(t & 3);
  double ql[3];
  load3(q, t, ql);
  const double qq[4] = {quat[4 * i], quat[4 * i + 1], quat[4 * i + 2], quat[4 * i + 3]};
  ts.commit(tab);
  double Rm[9], gB[3];
  quat_to_matrix(qq, Rm);
  const double gW[3] = {0.0, 0.0, -P.grav};
  irot(Rm, gW, gB);
  double F[3], J[9], Gq[3];
  leg_kinematics(LdsTab{tab + kTabPerLeg * leg}, ql, gB, F, J, Gq);
  if (!live) return;
  if (foot) { foot[3 * t] = F[0]; foot[3 * t + 1] = F[1]; foot[3 * t + 2] = F[2]; }
  if (jac) {
#pragma unroll
    for (int k = 0; k < 9; k++) jac[9 * t + k] = J[k];
  }
  if (grav) { grav[3 * t] = Gq[0]; grav[3 * t + 1] = Gq[1]; grav[3 * t + 2] = Gq[2]; }
}

// ---- analytic leg IK (row f4), one lane per (robot, leg) -------------------------------------------------
struct IkGeom { double g[3]; uint8_t config[4]; };

__global__ __launch_bounds__(64) void leg_ik_kernel(const DeviceParams *__restrict__ Pp, const IkGeom G,
                                                    const double *__restrict__ foot, const double *__restrict__ q_last,
                                                    int64_t B, double *__restrict__ q_out, uint8_t *__restrict__ ok) {
  const DeviceParams &P = *Pp;
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= 4 * B) return;
  const int leg = (int)(t & 3);
  // the hip frame (12 constants) and the inputs, all loads issued before the first use
  const double *tb = P.legtab + kTabPerLeg * leg;
  double hip[12];
#pragma unroll
  for (int k = 0; k < 9; k++) hip[k] = tb[kTabR0 + k];
#pragma unroll
  for (int k = 0; k < 3; k++) hip[9 + k] = tb[kTabXyz + k];
  double p[3], last[3] = {0.0, 0.0, 0.0};
  load3(foot, t, p);
  if (q_last) load3(q_last, t, last);
  struct HipTab { // kTabR0 + k -> hip[k], kTabXyz + k -> hip[9 + k]: the only entries the IK reads
    const double *h;
    __device__ __forceinline__ double operator[](int i) const { return i < kTabXyz ? h[i - kTabR0] : h[9 + i - kTabXyz]; }
  };
  double q[3];
  const bool good = leg_inverse_kinematics(HipTab{hip}, p, G.config[leg], G.g, q);
  // on failure the caller's previous joint positions are kept (quadruped_state.cpp:289-294)
#pragma unroll
  for (int k = 0; k < 3; k++) q_out[3 * t + k] = good ? q[k] : (q_last ? last[k] : q[k]);
  if (ok) ok[t] = good ? 1 : 0;
}

// ---- row a18: swing-leg torque, one lane per (robot, leg) -------------------------------------
struct SwingPtrs {
  const double *q, *qd, *qd_old, *tpos, *tvel, *q_id;
  const uint8_t *support;
};

__global__ __launch_bounds__(64) void swing_leg_kernel(const DeviceParams *__restrict__ Pp, const SwingParamsDev SP,
                                                       const SwingPtrs s, int64_t B, double *__restrict__ tau) {
  __shared__ double tab[4 * kTabPerLeg];
  const DeviceParams &P = *Pp;
  TabStage ts;
  ts.issue(P);
  const int64_t t0 = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = t0 < 4 * B;
  const int64_t t = live ? t0 : 4 * B - 1;
  const int leg = (int)(t & 3);
  double q[3], qd[3], qo[3], tp[3], tv[3], qi[3];
  load3(s.q, t, q); load3(s.qd, t, qd); load3(s.qd_old, t, qo); load3(s.tpos, t, tp); load3(s.tvel, t, tv);
  load3(s.q_id ? s.q_id : s.q, t, qi);
  const bool support = s.support[t] != 0;
  ts.commit(tab);
  double out[3] = {0.0, 0.0, 0.0};
  if (!support) swing_leg_torque(LdsTab{tab + kTabPerLeg * leg}, SP, qi, q, qd, qo, tp, tv, out);
  if (!live) return;
  tau[3 * t] = out[0]; tau[3 * t + 1] = out[1]; tau[3 * t + 2] = out[2];
}

// ---- swing branch of update(): PID / gravity compensation / swing torque per leg mode ---------------------
struct SwingBranchPtrs {
  const double *quat, *cmd;
  const uint8_t *mode;
  double *e_last, *e_int;
};

__global__ __launch_bounds__(64) void swing_branch_kernel(const DeviceParams *__restrict__ Pp, const SwingParamsDev SP,
                                                          const PidParamsDev pid, const SwingPtrs s,
                                                          const SwingBranchPtrs b, double period, int64_t B,
                                                          double *__restrict__ effort) {
  __shared__ double tab[4 * kTabPerLeg];
  const DeviceParams &P = *Pp;
  TabStage ts;
  ts.issue(P);
  const int64_t t0 = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = t0 < 4 * B;
  const int64_t t = live ? t0 : 4 * B - 1;
  const int64_t i = t >> 2;
  const int leg = (int)(t & 3);
  double q[3], qd[3], qo[3], tp[3], tv[3], cmd[3], qi[3], el[3], ei[3];
  load3(s.q, t, q); load3(s.qd, t, qd); load3(s.qd_old, t, qo); load3(s.tpos, t, tp); load3(s.tvel, t, tv);
  load3(b.cmd, t, cmd); load3(s.q_id ? s.q_id : s.q, t, qi); load3(b.e_last, t, el); load3(b.e_int, t, ei);
  const double quat[4] = {b.quat[4 * i], b.quat[4 * i + 1], b.quat[4 * i + 2], b.quat[4 * i + 3]};
  const int mode = (b.mode ? b.mode : s.support)[t];
  const bool support = s.support[t] != 0;
  PidLeg pl; // this leg's gains out of the kernel arguments, fetched with everything else
  pid_leg_of(pid, leg, pl);
  ts.commit(tab);
  if (support || !live) return; // support legs keep the clamped QP torque already in `effort` (:497-502)
  double out[3];
  swing_branch_leg(LdsTab{tab + kTabPerLeg * leg}, SP, pl, b.mode ? mode : 0, quat, qi, q, qd, qo, tp, tv, cmd, period, el,
                   ei, out);
#pragma unroll
  for (int k = 0; k < 3; k++) { effort[3 * t + k] = out[k]; b.e_last[3 * t + k] = el[k]; b.e_int[3 * t + k] = ei[k]; }
}

// ---- config 5: one pose-optimisation problem per lane, 16 problems per wavefront --------------
struct PosePtrs {
  const double *stance, *nominal, *polygon, *rcom, *maxlen, *pose;
  const uint8_t *mask;
  const int32_t *nverts;
};
constexpr int kPosePerWave = 16;

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

__global__ __launch_bounds__(64) void pose_sqp_kernel(const PoseParamsDev P, const PosePtrs s, int64_t B,
                                                      double *__restrict__ pose_out, int32_t *__restrict__ iters,
                                                      int32_t *__restrict__ status) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kPosePerWave + lane;
  // the problem data lives in LDS (one record per lane): it is read a few values at a time over the whole SQP
  // loop, and holding its 60 doubles in registers next to the QP's made the compiler spill each freshly loaded
  // value to scratch, one memory round trip after the other
  __shared__ PoseProblem pbs[kPosePerWave];
  if (lane >= kPosePerWave || i >= B) return;
  PoseProblem &pb = pbs[lane];
  double pose[7];
  load_pose_problem(P, s, i, pb, pose);
  LdsScratch scr{lds + lane, kPosePerWave};
  int it = 0;
  const int st = pose_sqp6(P, pb, scr, pose, &it); // register-resident inner QP (gi6_core.hpp)
#pragma unroll
  for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
  if (iters) iters[i] = it;
  status[i] = st;
}

// Lane-cooperative form (csrc/pose_coop.hpp): 16 lanes per problem, 4 problems per wavefront -- the default.
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
  if (lr == 0) { // one lane per problem fetches the record; the row reads it back from LDS
    double ps[7];
    load_pose_problem(P, s, i, pbs[row], ps);
#pragma unroll
    for (int a = 0; a < 7; a++) pose0[row][a] = ps[a];
  }
  __syncthreads();
  double pose[7];
#pragma unroll
  for (int a = 0; a < 7; a++) pose[a] = pose0[row][a];
  int it = 0;
  const int st = coop::pose_sqp_coop(P, pbs[row], live, rows + row * coop::kPoseCoopLdsDoubles, pose, it);
  if (lr == 0 && live) {
#pragma unroll
    for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
    if (iters) iters[i] = it;
    status[i] = st;
  }
}

// PoseOptimizationQP (position only) and PoseConstraintsChecker, same problem layout
__global__ __launch_bounds__(64) void pose_qp_kernel(const PoseParamsDev P, const PosePtrs s, int64_t B,
                                                     double *__restrict__ pose_out, int32_t *__restrict__ status) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kPosePerWave + lane;
  if (lane >= kPosePerWave || i >= B) return;
  PoseProblem pb;
  double pose[7];
  load_pose_problem(P, s, i, pb, pose);
  LdsScratch scr{lds + lane, kPosePerWave};
  const int st = pose_qp(P, pb, scr, pose);
#pragma unroll
  for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
  status[i] = st;
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
__global__ __launch_bounds__(64) void base_auto_pose_kernel(const PoseParamsDev P, const PosePtrs s,
                                                            const double *__restrict__ sfo_in,
                                                            const double *__restrict__ min_len, double leg_tol, int64_t B,
                                                            double *__restrict__ pose_out, int32_t *__restrict__ stage,
                                                            int32_t *__restrict__ iters, int32_t *__restrict__ status) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kPosePerWave + lane;
  __shared__ PoseProblem pbs[kPosePerWave]; // see pose_sqp_kernel
  if (lane >= kPosePerWave || i >= B) return;
  PoseProblem &pb = pbs[lane];
  double pose[7], sfo[4][3], mn[4];
  load_pose_problem(P, s, i, pb, pose);
  load_sfo(s, sfo_in, i, sfo);
#pragma unroll
  for (int k = 0; k < 4; k++) mn[k] = min_len ? min_len[4 * i + P.leg_order[k]] : 0.0;
  LdsScratch scr{lds + lane, kPosePerWave};
  int stg = 0, it = 0;
  const int st = base_auto_optimize_pose(P, pb, sfo, mn, leg_tol, scr, pose, &stg, &it);
#pragma unroll
  for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
  if (stage) stage[i] = stg;
  if (iters) iters[i] = it;
  status[i] = st;
}

// ---- leg state machine (row f2): one robot per lane, flags and a few doubles in, flags out ----------
struct LegStatePtrs {
  const uint8_t *support_leg, *is_footstep, *contact;
  const double *phase, *joint_position;
  int8_t *limb_state;
  uint8_t *store_flag;
  double *stored_joint_position, *joint_command, *foot_target;
  uint8_t *support;
  int8_t *code;
};

__global__ __launch_bounds__(256) void leg_state_kernel(const LegStatePtrs s, int index_quirk, int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  LegStateRobot r;
  // four flags per robot travel as one 32-bit word
  const uint32_t sup = *reinterpret_cast<const uint32_t *>(s.support_leg + 4 * i);
  const uint32_t fst = *reinterpret_cast<const uint32_t *>(s.is_footstep + 4 * i);
  const uint32_t con = *reinterpret_cast<const uint32_t *>(s.contact + 4 * i);
  const uint32_t lst = *reinterpret_cast<const uint32_t *>(s.limb_state + 4 * i);
  const uint32_t sto = *reinterpret_cast<const uint32_t *>(s.store_flag + 4 * i);
  const double2 p01 = *reinterpret_cast<const double2 *>(s.phase + 4 * i);
  const double2 p23 = *reinterpret_cast<const double2 *>(s.phase + 4 * i + 2);
  const double ph[4] = {p01.x, p01.y, p23.x, p23.y};
#pragma unroll
  for (int l = 0; l < 4; l++) {
    r.support_leg[l] = ((sup >> (8 * l)) & 0xFFu) != 0;
    r.is_footstep[l] = ((fst >> (8 * l)) & 0xFFu) != 0;
    r.contact[l] = ((con >> (8 * l)) & 0xFFu) != 0;
    r.limb_state[l] = (int)(int8_t)((lst >> (8 * l)) & 0xFFu);
    r.store_flag[l] = ((sto >> (8 * l)) & 0xFFu) != 0;
    r.phase[l] = ph[l];
  }
  // every array the tick may touch is fetched up front (independent 16-byte loads, one round trip); what
  // the state machine decides only selects which values are written back
  double jp[12], sj[12], ft[12];
  {
    const double2 *pj = reinterpret_cast<const double2 *>(s.joint_position + 12 * i);
    const double2 *ps = reinterpret_cast<const double2 *>(s.stored_joint_position + 12 * i);
    const double2 *pf = reinterpret_cast<const double2 *>(s.foot_target + 12 * i);
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double2 a = pj[k], b = ps[k], c = pf[k];
      jp[2 * k] = a.x; jp[2 * k + 1] = a.y; sj[2 * k] = b.x; sj[2 * k + 1] = b.y; ft[2 * k] = c.x; ft[2 * k + 1] = c.y;
    }
  }
  const uint32_t sup_i = *reinterpret_cast<const uint32_t *>(s.support + 4 * i);
  leg_state_machine(r, index_quirk != 0);
  uint32_t lst_o = 0, sto_o = 0, code_o = 0;
  uint32_t sup_o = sup_i;
#pragma unroll
  for (int l = 0; l < 4; l++) {
    lst_o |= (uint32_t)(uint8_t)(int8_t)r.limb_state[l] << (8 * l);
    sto_o |= (r.store_flag[l] ? 1u : 0u) << (8 * l);
    code_o |= (uint32_t)(uint8_t)(int8_t)r.code[l] << (8 * l);
    if (r.support_written[l]) sup_o = (sup_o & ~(0xFFu << (8 * l))) | ((r.support[l] ? 1u : 0u) << (8 * l));
    if (r.nudge_bumped[l]) { s.foot_target[12 * i + 3 * l] = ft[3 * l] - 0.005; s.foot_target[12 * i + 3 * l + 2] = ft[3 * l + 2] + 0.02; }
    if (r.nudge_late[l]) s.foot_target[12 * i + 3 * l + 2] = ft[3 * l + 2] - 0.01;
    if (r.capture[l]) {
#pragma unroll
      for (int k = 0; k < 3; k++) s.stored_joint_position[12 * i + 3 * l + k] = jp[3 * l + k];
    }
    if (r.hold[l]) {
#pragma unroll
      for (int k = 0; k < 3; k++) s.joint_command[12 * i + 3 * l + k] = sj[3 * l + k];
    }
  }
  *reinterpret_cast<uint32_t *>(s.limb_state + 4 * i) = lst_o;
  *reinterpret_cast<uint32_t *>(s.store_flag + 4 * i) = sto_o;
  *reinterpret_cast<uint32_t *>(s.support + 4 * i) = sup_o;
  *reinterpret_cast<uint32_t *>(s.code + 4 * i) = code_o;
}

// ---- free_gait_msgs/RobotState wire format -> SoA (row f2): one message per lane ---------------------
struct RobotStateOutPtrs {
  double *des_pos, *des_quat, *des_linvel, *des_angvel, *joint_command, *foot_position, *foot_velocity,
      *foot_acceleration, *surface_normal, *phase;
  uint8_t *support_leg, *leg_mode;
};

// Byte source in LDS: aligned 32-bit reads joined with v_alignbyte (fields sit at arbitrary byte offsets).
struct LdsBytes {
  static constexpr bool kOverread = true; // the staging window extends 16 bytes past the last message
  const uint32_t *w; // LDS, word-aligned base
  uint32_t shift;    // byte position of message offset 0 relative to w
  __device__ __forceinline__ uint32_t u32(uint32_t at) const {
    const uint32_t b = at + shift;
    const uint32_t lo = w[b >> 2], hi = w[(b >> 2) + 1];
    return __builtin_amdgcn_alignbyte(hi, lo, b & 3u);
  }
  __device__ __forceinline__ uint8_t u8(uint32_t at) const {
    const uint32_t b = at + shift;
    return (uint8_t)(w[b >> 2] >> (8 * (b & 3u)));
  }
  __device__ __forceinline__ double f64(uint32_t at) const {
    const uint32_t b = at + shift;
    const uint32_t w0 = w[b >> 2], w1 = w[(b >> 2) + 1], w2 = w[(b >> 2) + 2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, b & 3u), hi = __builtin_amdgcn_alignbyte(w2, w1, b & 3u);
    return __hiloint2double((int)hi, (int)lo);
  }
};

// Two-pass walk over a message staged in LDS.  Pass 1 follows only the length fields (the one true dependency
// chain: every string / array length decides where the next field starts) and notes where the wanted payload
// sits; pass 2 reads the payload at those anchors with independent loads.  Same results as robot_state_unpack
// (wire_core.hpp), which stays the reference implementation for the host build and the global-memory fallback.
constexpr int kTplMaxFields = 128; // length fields a layout template can hold (a reference message has ~95)
struct WireSkeleton {
  const LdsBytes &p;
  uint32_t pos, cap; // cap = len + 1; pos saturates there ("bad")
  uint32_t *log;     // global (position, value) pairs of the layout template, or NULL: only block 0's first message logs
  uint32_t nf;
  __device__ __forceinline__ void skip(uint32_t n) { pos = min(pos + min(n, cap), cap); }
  __device__ __forceinline__ uint32_t len_field() { // read a uint32 at pos, step over it
    const uint32_t at = min(pos, cap - 1);
    const uint32_t v = p.u32(at);
    if (log && nf < (uint32_t)kTplMaxFields) { log[2 * nf] = at; log[2 * nf + 1] = v; }
    nf++;
    skip(4);
    return v;
  }
  __device__ __forceinline__ void header() { skip(12); skip(len_field()); }
};

enum WireAnchor : int { // uint32 slots per message
  kAnJointPos = 0,      // [4] start of *_leg_joints.position data
  kAnOdomPose = 4,      // start of base_pose.pose.pose.position
  kAnModeName = 5,      // [4] start of *_leg_mode.name bytes
  kAnModeLen = 9,       // [4] its length
  kAnModeFlag = 13,     // [4] support_leg byte
  kAnModeNormal = 17,   // [4] surface_normal.vector
  kAnTarget = 21,       // [4][3] target_{position,velocity,acceleration}[0] payload
  kAnJointCnt = 33,     // [4] number of entries in *_leg_joints.position
  kAnCount = 37
};

// Pass 1.  Returns the status; nf = number of length fields met, end_pos = position after the last field.
__device__ __forceinline__ int wire_lds_skeleton(const LdsBytes &src, int64_t len64, uint32_t *an, uint32_t *log, uint32_t &nf,
                                                 uint32_t &end_pos) {
  nf = 0u; end_pos = 0u;
  if (len64 < 0 || len64 > 0x7FFFFFF0ll) return kWireTruncated;
  const uint32_t len = (uint32_t)len64;
  WireSkeleton c{src, 0u, len + 1u, log, 0u};
  bool missing = false;
  // ---- pass 1: skeleton
#pragma nounroll
  for (int l = 0; l < 4; l++) { // sensor_msgs/JointState
    c.header();
    uint32_t nn = c.len_field();
#pragma nounroll
    for (; nn > 0 && c.pos < c.cap; nn--) c.skip(c.len_field());
    const uint32_t np = c.len_field();
    missing = missing || np < 3;
    an[kAnJointPos + l] = c.pos;
    an[kAnJointCnt + l] = np;
    c.skip(np > 0x0FFFFFFFu ? 0xFFFFFFFFu : 8u * np);
    const uint32_t nv = c.len_field(); c.skip(nv > 0x0FFFFFFFu ? 0xFFFFFFFFu : 8u * nv);
    const uint32_t ne = c.len_field(); c.skip(ne > 0x0FFFFFFFu ? 0xFFFFFFFFu : 8u * ne);
  }
  c.header();                       // nav_msgs/Odometry
  c.skip(c.len_field());            // child_frame_id
  an[kAnOdomPose] = c.pos;
  c.skip(56 + 288 + 48 + 288);
#pragma nounroll
  for (int l = 0; l < 4; l++) {     // free_gait_msgs/LegMode
    const uint32_t n = c.len_field();
    an[kAnModeName + l] = c.pos;
    an[kAnModeLen + l] = n;
    c.skip(n);
    an[kAnModeFlag + l] = c.pos;
    c.skip(1 + 8 + 8);              // support_leg, duration, phase
    c.header();
    an[kAnModeNormal + l] = c.pos;
    c.skip(24 + 1);
  }
#pragma nounroll
  for (int l = 0; l < 4; l++) {     // free_gait_msgs/EndEffectorTarget
    c.skip(c.len_field());          // name
#pragma nounroll
    for (int arr = 0; arr < 4; arr++) {
      uint32_t n = c.len_field();
      if (arr < 3) missing = missing || n == 0;
#pragma nounroll
      for (uint32_t k = 0; k < n && c.pos < c.cap; k++) {
        c.header();
        if (k == 0 && arr < 3) an[kAnTarget + 3 * l + arr] = c.pos;
        c.skip(24);
      }
    }
    c.skip(8);                      // average_velocity
    c.header();
    c.skip(24 + 2);                 // surface_normal.vector, ignore_contact, ignore_for_pose_adaptation
  }
  nf = c.nf; end_pos = c.pos;
  if (c.pos >= c.cap) return kWireTruncated; // some field ran past the end: the record stays cleared
  return missing ? kWireMissingField : kWireOk;
}

// Pass 2: payload at the anchors, by the 16 lanes of the message's row.  The 77 doubles of a record are numbered in the
// order of RobotStateFields (des_pos 0-2, des_quat 3-6, des_linvel 7-9, des_angvel 10-12, joint_command 13-24,
// foot_position / velocity / acceleration 25-60, surface_normal 61-72, phase 73-76); lane lr takes entries lr, lr + 16, ...
// All of a lane's reads are issued before its first store (a store into the record would otherwise fence the reads:
// the record and the staged bytes are both LDS).  Lanes 0-3 also decode leg lr's mode name and support flag.
__device__ __forceinline__ void wire_lds_extract_row(const LdsBytes &src, RobotStateFields &f, const uint32_t *an, int lr) {
  double val[5];
  bool put[5];
#pragma unroll
  for (int t = 0; t < 5; t++) {
    const int e = lr + 16 * t;
    uint32_t slot = kAnOdomPose, off = 0u;
    bool ok = e < 77;
    if (e < 3) { off = 8u * e; }                                                   // des_pos
    else if (e < 7) { const int k = (e - 3 + 3) & 3; off = 24u + 8u * k; }          // des_quat (w,x,y,z) <- wire (x,y,z,w)
    else if (e < 13) { off = 56u + 288u + 8u * (e - 7); }                          // des_linvel, des_angvel
    else if (e < 25) { const int l = (e - 13) / 3, j = (e - 13) - 3 * l;           // joint_command
      slot = kAnJointPos + l; off = 8u * j; ok = (uint32_t)j < an[kAnJointCnt + l]; }
    else if (e < 61) { const int t9 = e - 25, arr = t9 / 12, r = t9 - 12 * arr, l = r / 3, j = r - 3 * l; // foot_* [arr][leg][j]
      slot = kAnTarget + 3 * l + arr; off = 8u * j; ok = an[slot] != 0u; }
    else if (e < 73) { const int l = (e - 61) / 3, j = (e - 61) - 3 * l; slot = kAnModeNormal + l; off = 8u * j; }
    else { slot = kAnModeFlag + (e - 73 < 4 ? e - 73 : 0); off = 9u; }               // phase
    put[t] = ok;
    val[t] = src.f64(an[ok ? slot : (uint32_t)kAnOdomPose] + (ok ? off : 0u));
  }
  int mode = kModeOther;
  uint8_t sup = 0;
  {
    const int l = lr & 3;
    const uint32_t nm = an[kAnModeName + l], nl = an[kAnModeLen + l];
    // "joint" 5, "leg_mode" 8, "cartesian" 9, "footstep" 8: compare 12 bytes read as three words against the literals
    const uint32_t w0 = src.u32(nm), w1 = src.u32(nm + 4), w2 = src.u32(nm + 8);
    if (nl == 5 && w0 == 0x6E696F6Au && (w1 & 0xFFu) == 0x74u) mode = kModeJoint;                    // "join" "t"
    else if (nl == 8 && w0 == 0x5F67656Cu && w1 == 0x65646F6Du) mode = kModeLegMode;                // "leg_" "mode"
    else if (nl == 9 && w0 == 0x74726163u && w1 == 0x61697365u && (w2 & 0xFFu) == 0x6Eu) mode = kModeCartesian; // "cart" "esia" "n"
    else if (nl == 8 && w0 == 0x746F6F66u && w1 == 0x70657473u) mode = kModeFootstep;               // "foot" "step"
    sup = src.u8(an[kAnModeFlag + l]) != 0;
  }
  double *rec = reinterpret_cast<double *>(&f);
#pragma unroll
  for (int t = 0; t < 5; t++)
    if (put[t]) rec[lr + 16 * t] = val[t];
  if (lr < 4) { f.leg_mode[lr] = (uint8_t)mode; f.support_leg[lr] = sup; }
}

constexpr int kWireMsgsPerBlock = 4;          // messages parsed per 64-lane block (one 16-lane row each)
constexpr int kWireLdsBytes = 32 * 1024;      // staging window; longer runs are parsed straight from global memory
// Layout template: the (position, value) of every length field of one well-formed message plus the anchors its walk
// produced.  A message whose length fields hold the template's values AT the template's positions has, by induction
// along the walk, exactly the template's layout -- so its anchors are known without walking.
constexpr uint32_t kTplMagic = 0x51574C54u;   // "TLWQ"
constexpr int kTplValid = 0, kTplEnd = 1, kTplMissing = 2, kTplFields = 3, kTplAnchors = 4, kTplPairs = kTplAnchors + kAnCount,
              kTplWords = kTplPairs + 2 * kTplMaxFields;

// One block = kWireMsgsPerBlock consecutive messages: the block copies their contiguous byte range into LDS with
// coalesced 16-byte loads; each message then belongs to one 16-lane row.  The row first checks the message against
// the layout template of the previous launch (tpl_in): its lanes compare the ~95 length fields in parallel.  On a hit
// the anchors are the template's; on a miss the row's first lane walks the length-prefixed fields (a chain of
// dependent LDS reads, ~15 us for a message).  The payload is then read at the anchors into a per-message record
// in LDS and the whole block writes the records out.  Block 0 leaves the template for the next launch in tpl_out
// (the layout of its first message if that one had to be walked, else the template it used).  Results never depend
// on the template, only the time does: streams from one publisher keep one layout.
__global__ __launch_bounds__(64) void robot_state_unpack_kernel(const uint8_t *__restrict__ messages,
                                                                const int64_t *__restrict__ offsets, int64_t B,
                                                                const RobotStateOutPtrs o, int32_t *__restrict__ status,
                                                                const uint32_t *__restrict__ tpl_in,
                                                                uint32_t *__restrict__ tpl_out) {
  extern __shared__ uint32_t wire_lds[];
  __shared__ RobotStateFields rec[kWireMsgsPerBlock];
  __shared__ uint32_t anchors[kWireMsgsPerBlock][kAnCount];
  __shared__ uint32_t tpl[kTplWords];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * kWireMsgsPerBlock;
  const int n = (int)((B - i0) < kWireMsgsPerBlock ? (B - i0) : kWireMsgsPerBlock);
  QL_STAMP(20);
  const int64_t a = offsets[i0], b = offsets[i0 + n];
  // the template's loads go out first, its LDS stores follow the staging loop
  uint32_t tplv[(kTplWords + 63) / 64];
#pragma unroll
  for (int j = 0; j < (kTplWords + 63) / 64; j++) tplv[j] = tpl_in[min(tid + 64 * j, kTplWords - 1)];
  const uintptr_t src = (uintptr_t)(messages + a);
  const uintptr_t src_al = src & ~(uintptr_t)15;
  const int64_t lead = (int64_t)(src - src_al), nbytes = lead + (b - a);
  const bool staged = nbytes + 16 <= kWireLdsBytes; // +16: u32 / f64 reads may touch the next two words
  if (staged) {
    const int64_t full = nbytes >> 4;
    const uint4 *g = (const uint4 *)src_al;
    uint4 *l4 = (uint4 *)wire_lds;
    // sixteen 16-byte loads in flight per lane (a 16 KB window: four typical messages) before the first LDS store:
    // one DRAM round trip for the block instead of one per kilobyte
    for (int64_t k0 = tid; k0 - tid < full; k0 += 64 * 16) {
      const int64_t last = full - 1;
      const uint4 v0 = g[min(k0 + 0, last)], v1 = g[min(k0 + 64, last)], v2 = g[min(k0 + 128, last)], v3 = g[min(k0 + 192, last)], v4 = g[min(k0 + 256, last)], v5 = g[min(k0 + 320, last)], v6 = g[min(k0 + 384, last)], v7 = g[min(k0 + 448, last)], v8 = g[min(k0 + 512, last)], v9 = g[min(k0 + 576, last)], v10 = g[min(k0 + 640, last)], v11 = g[min(k0 + 704, last)], v12 = g[min(k0 + 768, last)], v13 = g[min(k0 + 832, last)], v14 = g[min(k0 + 896, last)], v15 = g[min(k0 + 960, last)];
      if (k0 + 0 < full) l4[k0 + 0] = v0;
      if (k0 + 64 < full) l4[k0 + 64] = v1;
      if (k0 + 128 < full) l4[k0 + 128] = v2;
      if (k0 + 192 < full) l4[k0 + 192] = v3;
      if (k0 + 256 < full) l4[k0 + 256] = v4;
      if (k0 + 320 < full) l4[k0 + 320] = v5;
      if (k0 + 384 < full) l4[k0 + 384] = v6;
      if (k0 + 448 < full) l4[k0 + 448] = v7;
      if (k0 + 512 < full) l4[k0 + 512] = v8;
      if (k0 + 576 < full) l4[k0 + 576] = v9;
      if (k0 + 640 < full) l4[k0 + 640] = v10;
      if (k0 + 704 < full) l4[k0 + 704] = v11;
      if (k0 + 768 < full) l4[k0 + 768] = v12;
      if (k0 + 832 < full) l4[k0 + 832] = v13;
      if (k0 + 896 < full) l4[k0 + 896] = v14;
      if (k0 + 960 < full) l4[k0 + 960] = v15;
    }
    const int64_t tail0 = full << 4;                 // last partial chunk byte by byte: never read past the blob
    if (tid < nbytes - tail0) ((uint8_t *)wire_lds)[tail0 + tid] = ((const uint8_t *)src_al)[tail0 + tid];
  }
#pragma unroll
  for (int j = 0; j < (kTplWords + 63) / 64; j++)
    if (tid + 64 * j < kTplWords) tpl[tid + 64 * j] = tplv[j];
  // clear the records (fields a malformed message never reaches read as zero)
  for (int w = tid; w < (int)(sizeof(rec) / 4); w += 64) ((uint32_t *)rec)[w] = 0u;
  for (int w = tid; w < kWireMsgsPerBlock * kAnCount; w += 64) (&anchors[0][0])[w] = 0u;
  __syncthreads();
  QL_STAMP(21);
  const int row = tid >> 4, lr = tid & 15;
  const bool mine = row < n;
  const int64_t ma = offsets[i0 + (mine ? row : 0)], mb = offsets[i0 + (mine ? row : 0) + 1];
  const bool sane = mine && !(ma < a || mb > b || mb < ma); // offsets not ascending: nothing to parse
  const LdsBytes msg{wire_lds, (uint32_t)(lead + (ma - a))};
  // ---- template check, 16 lanes per message
  const uint32_t tnf = tpl[kTplFields];
  bool same = sane && staged && tpl[kTplValid] == kTplMagic && tnf <= (uint32_t)kTplMaxFields &&
              (mb - ma) <= 0x7FFFFFF0ll && (uint64_t)(mb - ma) >= (uint64_t)tpl[kTplEnd];
  {
    // eight length fields per lane, all reads independent (positions inside the message: <= end <= length)
    uint32_t at[kTplMaxFields / 16], want[kTplMaxFields / 16], got[kTplMaxFields / 16];
#pragma unroll
    for (int t = 0; t < kTplMaxFields / 16; t++) {
      const uint32_t k = lr + 16u * t;
      const bool on = same && k < tnf;
      at[t] = on ? tpl[kTplPairs + 2 * k] : 0u;
      want[t] = on ? tpl[kTplPairs + 2 * k + 1] : 0u;
    }
#pragma unroll
    for (int t = 0; t < kTplMaxFields / 16; t++) got[t] = msg.u32(same ? at[t] : 0u);
#pragma unroll
    for (int t = 0; t < kTplMaxFields / 16; t++) same = same && (lr + 16u * t >= tnf || got[t] == want[t]);
  }
  const bool hit = ((unsigned)(__ballot(same) >> (tid & 48)) & 0xFFFFu) == 0xFFFFu;
  if (hit)
    for (int k = lr; k < kAnCount; k += 16) anchors[row][k] = tpl[kTplAnchors + k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  QL_STAMP(22);
  uint32_t nf = 0u, end_pos = 0u;
  int st = kWireTruncated;
  const bool logger = blockIdx.x == 0 && row == 0; // this message's layout becomes the next launch's template
  if (lr == 0 && mine) {
    if (!sane) st = kWireTruncated;
    else if (hit) st = tpl[kTplMissing] ? kWireMissingField : kWireOk;
    else if (staged) st = wire_lds_skeleton(msg, mb - ma, anchors[row], logger ? tpl_out + kTplPairs : nullptr, nf, end_pos);
    else st = robot_state_unpack(PlainBytes{messages + ma}, mb - ma, rec[row]);
    status[i0 + row] = st;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const int st_row = __shfl(st, 0, 16);
  if (mine && sane && staged && st_row != kWireTruncated) wire_lds_extract_row(msg, rec[row], anchors[row], lr);
  QL_STAMP(23);
  // ---- the template for the next launch (block 0, first message)
  if (logger) {
    const uint32_t nf0 = __shfl(nf, 0, 16), end0 = __shfl(end_pos, 0, 16);
    if (hit) {
      for (int k = lr; k < kTplWords; k += 16) tpl_out[k] = tpl[k]; // still in force
    } else {
      // the walk has already left its (position, value) pairs in tpl_out; valid only if the message was well-formed
      const bool good = sane && staged && st_row != kWireTruncated && nf0 <= (uint32_t)kTplMaxFields;
      if (lr == 0) {
        tpl_out[kTplValid] = good ? kTplMagic : 0u; tpl_out[kTplEnd] = end0;
        tpl_out[kTplMissing] = st_row == kWireMissingField ? 1u : 0u; tpl_out[kTplFields] = nf0;
      }
      for (int k = lr; k < kAnCount; k += 16) tpl_out[kTplAnchors + k] = anchors[0][k];
    }
  }
  __syncthreads();
  QL_STAMP(24);
  // write-out: message m's k doubles of each field are contiguous in the output arrays
  const auto put = [&](double *dst, int width, size_t field_off) {
    if (!dst) return;
    for (int e = tid; e < n * width; e += 64) {
      const int m = e / width, k2 = e - m * width;
      dst[(int64_t)width * i0 + e] = ((const double *)((const char *)&rec[m] + field_off))[k2];
    }
  };
  put(o.des_pos, 3, offsetof(RobotStateFields, des_pos)); put(o.des_quat, 4, offsetof(RobotStateFields, des_quat));
  put(o.des_linvel, 3, offsetof(RobotStateFields, des_linvel)); put(o.des_angvel, 3, offsetof(RobotStateFields, des_angvel));
  put(o.joint_command, 12, offsetof(RobotStateFields, joint_command));
  put(o.foot_position, 12, offsetof(RobotStateFields, foot_position));
  put(o.foot_velocity, 12, offsetof(RobotStateFields, foot_velocity));
  put(o.foot_acceleration, 12, offsetof(RobotStateFields, foot_acceleration));
  put(o.surface_normal, 12, offsetof(RobotStateFields, surface_normal)); put(o.phase, 4, offsetof(RobotStateFields, phase));
  if (tid < 4 * n) {
    const int m = tid >> 2, l = tid & 3;
    if (o.support_leg) o.support_leg[4 * i0 + tid] = rec[m].support_leg[l];
    if (o.leg_mode) o.leg_mode[4 * i0 + tid] = rec[m].leg_mode[l];
  }
  QL_STAMP(25);
}

// ---- dense QP batch: one problem per lane, 8 problems per wavefront, arrays in LDS ------------
typedef GiLayout<12, 2, 24> QpGi;
constexpr int kQpPerWave = 8;

__global__ __launch_bounds__(64) void qp_solve_kernel(int n, int p, int m, const double *__restrict__ G,
                                                      const double *__restrict__ g0, const double *__restrict__ CE,
                                                      const double *__restrict__ ce0, const double *__restrict__ CI,
                                                      const double *__restrict__ ci0, int64_t B,
                                                      double *__restrict__ x, double *__restrict__ obj,
                                                      int32_t *__restrict__ status) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kQpPerWave + lane;
  if (lane >= kQpPerWave || i >= B) return;
  LdsScratch s{lds + lane, kQpPerWave};
  for (int k = 0; k < n * n; k++) s.at(QpGi::G + k) = G[(size_t)i * n * n + k];
  for (int k = 0; k < n; k++) s.at(QpGi::G0 + k) = g0[(size_t)i * n + k];
  for (int k = 0; k < n * p; k++) s.at(QpGi::CE + k) = CE[(size_t)i * n * p + k];
  for (int k = 0; k < p; k++) s.at(QpGi::CE0 + k) = ce0[(size_t)i * p + k];
  for (int k = 0; k < n * m; k++) s.at(QpGi::CI + k) = CI[(size_t)i * n * m + k];
  for (int k = 0; k < m; k++) s.at(QpGi::CI0 + k) = ci0[(size_t)i * m + k];
  double f;
  const int st = gi_solve<12, 2, 24>(s, n, p, m, &f, nullptr);
  for (int k = 0; k < n; k++) x[(size_t)i * n + k] = s.at(QpGi::X + k);
  if (obj) obj[i] = f;
  status[i] = st;
}

// Lane-cooperative dense QP batch (csrc/qp_coop.hpp): 16 lanes per problem, 4 problems per wavefront.
// N = 6 for n <= 6, N = 12 otherwise; KC = 2 inequalities per lane for m <= 24, 3 for m <= 48; at most one equality
// column (two go to qp_solve_kernel).
template <int N, int KC>
__global__ __launch_bounds__(64) void qp_coop_kernel(int n, int p, int m, const double *__restrict__ G,
                                                     const double *__restrict__ g0, const double *__restrict__ CE,
                                                     const double *__restrict__ ce0, const double *__restrict__ CI,
                                                     const double *__restrict__ ci0, int64_t B, double *__restrict__ x,
                                                     double *__restrict__ obj, int32_t *__restrict__ status) {
  typedef coop::QpCoopLds<N, KC> L;
  __shared__ double rows[coop::kQpCoopRows * L::kTotal];
  const int row = threadIdx.x >> 4, lr = threadIdx.x & 15;
  int64_t i = (int64_t)blockIdx.x * coop::kQpCoopRows + row;
  const bool live = i < B;
  if (!live) i = B - 1;
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
#pragma unroll
  for (int s = 0; s < KC; s++) b[s] = v[s] ? b[s] : 0.0;
  double xo, fo;
  const int st = coop::qp_coop_impl<N, KC>(Gm, gl, n, n, m, p > 0, ne, e0, a, b, v, !live, rows + row * L::kTotal, xo, fo);
  if (live) {
    if (var) x[(size_t)i * n + lr] = xo;
    if (lr == 0) {
      if (obj) obj[i] = fo;
      status[i] = st;
    }
  }
}

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
  for (int e = lr; e < kWbStage; e += 16) ob[e] = 0.0;

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
  double tau = 0.0, gb[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if constexpr (kHJ) wb_inverse_dynamics(W, L, c, V0, A0, c < 3 ? in.qdj : 0.0, 0.0, tau, gb);
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
    for (int e = threadIdx.x; e < 324 * nrob; e += 64) Mo[r0 * 324 + e] = outb[e]; // kWbStage == 324: robots are contiguous
    __syncthreads();
    if (kHJ && (ho || Jo))
      for (int e = lr; e < 18 + 216; e += 16) ob[e] = 0.0;
  }
  // ---- pass 2: bias forces and the contact Jacobian
  if (kHJ && (ho || Jo)) {
    static_for<6>([&](auto E) { constexpr int e = E; ob[kWbH + e] = gb[e]; });
    if (c < 3) {
      const int j = 6 + 3 * leg + c;
      ob[kWbH + j] = tau;
      double d[3] = {L.pf[0] - L.p[0], L.pf[1] - L.p[1], L.pf[2] - L.p[2]}, col[3];
      cross3(L.z, d, col);
#pragma unroll
      for (int a = 0; a < 3; a++) ob[kWbJc + 18 * (3 * leg + a) + j] = col[a];
    } else {
      // the foot lane writes [1 , -[r]x] of its leg's three rows
      double *jr = ob + kWbJc + 18 * 3 * leg;
      jr[0] = 1.0; jr[18 + 1] = 1.0; jr[36 + 2] = 1.0;
      jr[4] = L.pf[2]; jr[5] = -L.pf[1];            // -[r]x
      jr[18 + 3] = -L.pf[2]; jr[18 + 5] = L.pf[0];
      jr[36 + 3] = L.pf[1]; jr[36 + 4] = -L.pf[0];
    }
    __syncthreads();
    if (ho)
      for (int e = threadIdx.x; e < 18 * nrob; e += 64) ho[r0 * 18 + e] = outb[kWbStage * (e / 18) + kWbH + e % 18];
    if (Jo)
      for (int e = threadIdx.x; e < 216 * nrob; e += 64) Jo[r0 * 216 + e] = outb[kWbStage * (e / 216) + kWbJc + e % 216];
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

// ---- the whole tick: a known leg-mode name replaces the mode in force, anything else leaves it (:876-964) ----------
__global__ void tick_leg_mode_kernel(const uint8_t *__restrict__ msg_mode, uint8_t *__restrict__ leg_mode,
                                     uint8_t *__restrict__ is_footstep, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint8_t m = msg_mode[t];
  const uint8_t cur = m != kModeOther ? m : leg_mode[t];
  leg_mode[t] = cur;
  is_footstep[t] = cur == kModeFootstep ? 1 : 0;
}

} // namespace

// ------------------------------------------------------------------ C-ABI ---

struct qlamd_context {
  int device;
  DeviceParams params;
  DeviceParams *d_params; // device copy, read through scalar loads
  int rpw_override;
  int num_cu;
  double base_m, base_h[3], base_I[6]; // base_link about the base origin (whole-body entries)
  void *tick_ws;       // intermediates of qlamd_full_tick_batch (grown on demand)
  size_t tick_ws_bytes;
  uint32_t *wire_tpl;  // two layout templates of robot_state_unpack_kernel (read one, write the other), or NULL
  int wire_flip;
  // HOST-memory mode staging (grown on demand)
  void *ws;
  size_t ws_bytes;
  void *pinned;        // page-locked mirror of the head of ws, for small host-buffer calls (one copy each way)
  size_t pinned_bytes;
};

namespace {

int pick_rpw(const qlamd_context *ctx, int64_t batch) {
  // The lane-cooperative kernel wins at every batch size measured (1 K ... 1 M robots, static and
  // trot: tools/batch_sweep.py); the one-lane-per-robot kernels stay selectable as an independent
  // second implementation (different QP linear algebra) for cross-checks.
  (void)batch;
  return ctx->rpw_override ? ctx->rpw_override : 4;
}

template <int RPW>
hipError_t launch_balance(const qlamd_context *ctx, const StatePtrs &s, int64_t B, double *tau, double *grf,
                          int32_t *status, hipStream_t st) {
  const size_t lds = ((size_t)RPW * kScratchDoubles + 4 * kTabPerLeg) * sizeof(double);
  const unsigned grid = (unsigned)((B + RPW - 1) / RPW);
  if (s.normals) {
    auto k = balance_step_kernel<RPW, true>;
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, ctx->d_params, s, B, tau, grf, status);
  } else {
    auto k = balance_step_kernel<RPW, false>;
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, ctx->d_params, s, B, tau, grf, status);
  }
  return hipGetLastError();
}

int ensure_ws(qlamd_context *ctx, size_t bytes) {
  if (ctx->ws_bytes >= bytes) return QLAMD_OK;
  if (ctx->ws) (void)hipFree(ctx->ws);
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  if (hipMalloc(&ctx->ws, bytes) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
  ctx->ws_bytes = bytes;
  return QLAMD_OK;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int ensure_pinned(qlamd_context *ctx, size_t bytes) {
  if (ctx->pinned_bytes >= bytes) return QLAMD_OK;
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  ctx->pinned = nullptr;
  ctx->pinned_bytes = 0;
  if (hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
  ctx->pinned_bytes = bytes;
  return QLAMD_OK;
}
constexpr size_t kSmallHostCall = 256 * 1024; // below this a host-buffer call goes through one pinned slab

// Host-buffer calls: the listed arrays are laid out in the context workspace, inputs copied up front,
// outputs copied back (and the stream synchronised) by finish().  Calls whose arrays total at most
// kSmallHostCall bytes go through the context's pinned slab: one copy up (the span of the inputs) and
// one copy down (the span of the outputs) instead of one pageable copy per array.
struct Staged {
  struct Item { void *host; size_t bytes; bool in, out; size_t off; };
  Item items[24];
  int n = 0;
  char *base = nullptr;
  char *slab = nullptr; // pinned mirror of the workspace for small calls
  int add(const void *host, size_t bytes, bool in, bool out) {
    items[n] = Item{const_cast<void *>(host), host ? bytes : 0, in, out, 0};
    return n++;
  }
  void span(bool want_out, size_t *lo, size_t *hi) const {
    *lo = ~(size_t)0; *hi = 0;
    for (int k = 0; k < n; k++) {
      if (!items[k].bytes || !(want_out ? items[k].out : items[k].in)) continue;
      if (items[k].off < *lo) *lo = items[k].off;
      if (items[k].off + items[k].bytes > *hi) *hi = items[k].off + items[k].bytes;
    }
  }
  int upload(qlamd_context *ctx, hipStream_t st) {
    size_t total = 0;
    for (int k = 0; k < n; k++) { items[k].off = total; total += align256(items[k].bytes); }
    int rc = ensure_ws(ctx, total ? total : 256);
    if (rc != QLAMD_OK) return rc;
    base = (char *)ctx->ws;
    if (total && total <= kSmallHostCall) {
      rc = ensure_pinned(ctx, kSmallHostCall);
      if (rc != QLAMD_OK) return rc;
      slab = (char *)ctx->pinned;
      for (int k = 0; k < n; k++)
        if (items[k].in && items[k].bytes) memcpy(slab + items[k].off, items[k].host, items[k].bytes);
      size_t lo, hi;
      span(false, &lo, &hi);
      if (hi > lo && hipMemcpyAsync(base + lo, slab + lo, hi - lo, hipMemcpyHostToDevice, st) != hipSuccess)
        return QLAMD_ERR_HIP;
      return QLAMD_OK;
    }
    for (int k = 0; k < n; k++)
      if (items[k].in && items[k].bytes &&
          hipMemcpyAsync(base + items[k].off, items[k].host, items[k].bytes, hipMemcpyHostToDevice, st) != hipSuccess)
        return QLAMD_ERR_HIP;
    return QLAMD_OK;
  }
  template <class T> T *dev(int k) const { return items[k].host ? (T *)(base + items[k].off) : nullptr; }
  int finish(hipStream_t st) {
    if (slab) {
      size_t lo, hi;
      span(true, &lo, &hi);
      if (hi > lo && hipMemcpyAsync(slab + lo, base + lo, hi - lo, hipMemcpyDeviceToHost, st) != hipSuccess)
        return QLAMD_ERR_HIP;
      if (hipStreamSynchronize(st) != hipSuccess) return QLAMD_ERR_HIP;
      for (int k = 0; k < n; k++)
        if (items[k].out && items[k].bytes) memcpy(items[k].host, slab + items[k].off, items[k].bytes);
      return QLAMD_OK;
    }
    for (int k = 0; k < n; k++)
      if (items[k].out && items[k].bytes &&
          hipMemcpyAsync(items[k].host, base + items[k].off, items[k].bytes, hipMemcpyDeviceToHost, st) != hipSuccess)
        return QLAMD_ERR_HIP;
    return hipStreamSynchronize(st) == hipSuccess ? QLAMD_OK : QLAMD_ERR_HIP;
  }
};

} // namespace

extern "C" {

void qlamd_balance_default_params(qlamd_balance_params *p) { if (p) default_balance_params(p); }
void qlamd_default_robot_model(qlamd_robot_model *m) { if (m) default_robot_model(m); }
int qlamd_version(void) { return QLAMD_VERSION_MAJOR * 1000 + QLAMD_VERSION_MINOR; }

void qlamd_swing_default_params(qlamd_swing_params *p) {
  if (!p) return;
  for (int i = 0; i < 3; i++) { p->kp[i] = 300.0; p->kd[i] = 20.0; } // controller_gains.yaml:42-51
  p->period = 0.0025;      // balance_controller_manager.cpp:48
  p->accel_window = 10.0;  // model_test_header.cpp:418
  p->accel_scale = 0.5;    // model_test_header.cpp:460
  p->gravity = 9.81;
}

int qlamd_swing_leg_torque_batch(qlamd_context *ctx, const qlamd_swing_params *params, const qlamd_swing_batch *in,
                                 int64_t batch, double *joint_effort, int memory, void *stream) {
  if (!ctx || !in || batch < 0 || !joint_effort) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!params) return QLAMD_ERR_NOT_LOADED;
  if (!in->joint_position || !in->joint_velocity || !in->joint_velocity_oldest || !in->target_foot_position ||
      !in->target_foot_velocity || !in->support_leg)
    return QLAMD_ERR_INVALID_ARGUMENT;
  if (!(params->period > 0.0) || !(params->accel_window > 0.0)) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)batch;
  SwingParamsDev SP;
  for (int i = 0; i < 3; i++) { SP.kp[i] = params->kp[i]; SP.kd[i] = params->kd[i]; }
  SP.period = params->period; SP.accel_window = params->accel_window; SP.accel_scale = params->accel_scale;
  SP.gravity = params->gravity;
  SwingPtrs s{in->joint_position, in->joint_velocity, in->joint_velocity_oldest, in->target_foot_position,
              in->target_foot_velocity, in->id_joint_position, in->support_leg};
  double *d_tau = joint_effort;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    sg.add(in->joint_position, B * 96, true, false);
    sg.add(in->joint_velocity, B * 96, true, false);
    sg.add(in->joint_velocity_oldest, B * 96, true, false);
    sg.add(in->target_foot_position, B * 96, true, false);
    sg.add(in->target_foot_velocity, B * 96, true, false);
    sg.add(in->id_joint_position, B * 96, true, false);
    sg.add(in->support_leg, B * 4, true, false);
    sg.add(joint_effort, B * 96, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    s = SwingPtrs{sg.dev<const double>(0), sg.dev<const double>(1), sg.dev<const double>(2), sg.dev<const double>(3),
                  sg.dev<const double>(4), sg.dev<const double>(5), sg.dev<const uint8_t>(6)};
    d_tau = sg.dev<double>(7);
  }
  const unsigned grid = (unsigned)((4 * batch + 63) / 64);
  hipLaunchKernelGGL(swing_leg_kernel, dim3(grid), dim3(64), 0, st, ctx->d_params, SP, s, batch, d_tau);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

void qlamd_joint_pid_default_params(qlamd_joint_pid_params *p) {
  if (!p) return;
  for (int j = 0; j < 12; j++) { // balance_controller/config/control.yaml:18-29; limits quadruped_model.urdf:53-57
    p->p[j] = 300.0; p->i[j] = 0.01; p->d[j] = 3.0;
    p->i_max[j] = 0.0; p->i_min[j] = 0.0;
    p->lower[j] = -3.0; p->upper[j] = 3.0;
  }
  p->antiwindup = 0;
}

int qlamd_swing_branch_batch(qlamd_context *ctx, const qlamd_swing_params *params, const qlamd_joint_pid_params *pid,
                             const qlamd_swing_batch *in, const qlamd_swing_branch_extra *extra, double period,
                             int64_t batch, double *joint_effort, int memory, void *stream) {
  if (!ctx || !in || !extra || batch < 0 || !joint_effort) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!params || !pid) return QLAMD_ERR_NOT_LOADED;
  if (!in->joint_position || !in->joint_velocity || !in->joint_velocity_oldest || !in->target_foot_position ||
      !in->target_foot_velocity || !in->support_leg || !extra->base_orientation || !extra->joint_command ||
      !extra->pid_error_last || !extra->pid_error_integral)
    return QLAMD_ERR_INVALID_ARGUMENT;
  if (!(params->period > 0.0) || !(params->accel_window > 0.0)) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)batch;
  SwingParamsDev SP;
  for (int i = 0; i < 3; i++) { SP.kp[i] = params->kp[i]; SP.kd[i] = params->kd[i]; }
  SP.period = params->period; SP.accel_window = params->accel_window; SP.accel_scale = params->accel_scale;
  SP.gravity = params->gravity;
  PidParamsDev PD;
  memcpy(PD.p, pid->p, sizeof(PD.p)); memcpy(PD.i, pid->i, sizeof(PD.i)); memcpy(PD.d, pid->d, sizeof(PD.d));
  memcpy(PD.i_max, pid->i_max, sizeof(PD.i_max)); memcpy(PD.i_min, pid->i_min, sizeof(PD.i_min));
  memcpy(PD.lower, pid->lower, sizeof(PD.lower)); memcpy(PD.upper, pid->upper, sizeof(PD.upper));
  PD.antiwindup = pid->antiwindup;
  SwingPtrs s{in->joint_position, in->joint_velocity, in->joint_velocity_oldest, in->target_foot_position,
              in->target_foot_velocity, in->id_joint_position, in->support_leg};
  SwingBranchPtrs sb{extra->base_orientation, extra->joint_command, extra->leg_mode, extra->pid_error_last,
                     extra->pid_error_integral};
  double *d_eff = joint_effort;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    const int a0 = sg.add(in->joint_position, B * 96, true, false), a1 = sg.add(in->joint_velocity, B * 96, true, false);
    const int a2 = sg.add(in->joint_velocity_oldest, B * 96, true, false);
    const int a3 = sg.add(in->target_foot_position, B * 96, true, false);
    const int a4 = sg.add(in->target_foot_velocity, B * 96, true, false);
    const int a5 = sg.add(in->id_joint_position, B * 96, true, false), a6 = sg.add(in->support_leg, B * 4, true, false);
    const int b0 = sg.add(extra->base_orientation, B * 32, true, false), b1 = sg.add(extra->joint_command, B * 96, true, false);
    const int b2 = sg.add(extra->leg_mode, B * 4, true, false);
    const int b3 = sg.add(extra->pid_error_last, B * 96, true, true), b4 = sg.add(extra->pid_error_integral, B * 96, true, true);
    const int e0 = sg.add(joint_effort, B * 96, true, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    s = SwingPtrs{sg.dev<const double>(a0), sg.dev<const double>(a1), sg.dev<const double>(a2), sg.dev<const double>(a3),
                  sg.dev<const double>(a4), sg.dev<const double>(a5), sg.dev<const uint8_t>(a6)};
    sb = SwingBranchPtrs{sg.dev<const double>(b0), sg.dev<const double>(b1), sg.dev<const uint8_t>(b2), sg.dev<double>(b3),
                         sg.dev<double>(b4)};
    d_eff = sg.dev<double>(e0);
  }
  hipLaunchKernelGGL(swing_branch_kernel, dim3((unsigned)((4 * batch + 63) / 64)), dim3(64), 0, st, ctx->d_params, SP, PD,
                     s, sb, period, batch, d_eff);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  return memory == QLAMD_MEM_HOST ? sg.finish(st) : QLAMD_OK;
}

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
  const unsigned grid = (unsigned)((batch + kPosePerWave - 1) / kPosePerWave);
  const size_t lds6 = (size_t)kPosePerWave * Gi6Layout::kTotal * sizeof(double);
  const size_t lds3 = (size_t)kPosePerWave * PoseQpGi::kTotal * sizeof(double);
  switch (mode) {
    case kPoseSqp:
      if (getenv("QLAMD_POSE_ONE_LANE")) // the one-lane-per-problem form, kept as a second implementation
        hipLaunchKernelGGL(pose_sqp_kernel, dim3(grid), dim3(64), lds6, st, P, s, batch, d_out, d_it, d_st);
      else
        hipLaunchKernelGGL(pose_sqp_coop_kernel,
                           dim3((unsigned)((batch + coop::kPoseCoopRows - 1) / coop::kPoseCoopRows)), dim3(64), 0, st, P, s,
                           batch, d_out, d_it, d_st);
      break;
    case kPoseQp:
      hipLaunchKernelGGL(pose_qp_kernel, dim3(grid), dim3(64), lds3, st, P, s, batch, d_out, d_st);
      break;
    case kPoseCheck:
      hipLaunchKernelGGL(pose_check_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, P, s, d_min,
                         call.leg_tol, batch, d_ok);
      break;
    case kPoseGeometric:
      hipLaunchKernelGGL(pose_geometric_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, P, s, d_sfo, batch,
                         d_out);
      break;
    case kPoseBaseAuto:
      hipLaunchKernelGGL(base_auto_pose_kernel, dim3(grid), dim3(64), lds6 > lds3 ? lds6 : lds3, st, P, s, d_sfo, d_min,
                         call.leg_tol, batch, d_out, d_stage, d_it, d_st);
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

int qlamd_leg_state_machine_batch(qlamd_context *ctx, const qlamd_leg_state_batch *io, int index_quirk, int64_t batch,
                                  int memory, void *stream) {
  if (!ctx || !io || batch < 0) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!io->support_leg || !io->phase || !io->is_footstep || !io->contact || !io->joint_position || !io->limb_state ||
      !io->store_flag || !io->stored_joint_position || !io->joint_command || !io->foot_target || !io->support ||
      !io->leg_state_code)
    return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)batch;
  LegStatePtrs s{io->support_leg, io->is_footstep, io->contact, io->phase, io->joint_position, io->limb_state,
                 io->store_flag, io->stored_joint_position, io->joint_command, io->foot_target, io->support,
                 io->leg_state_code};
  // host staging: every array goes up except leg_state_code; the in/out and out arrays come back
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    sg.add(io->support_leg, B * 4, true, false);
    sg.add(io->is_footstep, B * 4, true, false);
    sg.add(io->contact, B * 4, true, false);
    sg.add(io->phase, B * 32, true, false);
    sg.add(io->joint_position, B * 96, true, false);
    sg.add(io->limb_state, B * 4, true, true);
    sg.add(io->store_flag, B * 4, true, true);
    sg.add(io->stored_joint_position, B * 96, true, true);
    sg.add(io->joint_command, B * 96, true, true);
    sg.add(io->foot_target, B * 96, true, true);
    sg.add(io->support, B * 4, true, true);
    sg.add(io->leg_state_code, B * 4, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    s = LegStatePtrs{sg.dev<const uint8_t>(0), sg.dev<const uint8_t>(1), sg.dev<const uint8_t>(2), sg.dev<const double>(3),
                     sg.dev<const double>(4), sg.dev<int8_t>(5), sg.dev<uint8_t>(6), sg.dev<double>(7), sg.dev<double>(8),
                     sg.dev<double>(9), sg.dev<uint8_t>(10), sg.dev<int8_t>(11)};
  }
  hipLaunchKernelGGL(leg_state_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, s, index_quirk, batch);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

int qlamd_robot_state_unpack_batch(qlamd_context *ctx, const uint8_t *messages, const int64_t *offsets, int64_t batch,
                                   const qlamd_robot_state_fields *out, int32_t *status, int memory, void *stream) {
  if (!ctx || !messages || !offsets || !out || !status || batch < 0) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)batch;
  enum { kD = 10 };
  const int width[kD] = {3, 4, 3, 3, 12, 12, 12, 12, 12, 4};
  double *hostd[kD] = {out->des_pos, out->des_quat, out->des_linvel, out->des_angvel, out->joint_command,
                       out->foot_position, out->foot_velocity, out->foot_acceleration, out->surface_normal, out->phase};
  RobotStateOutPtrs o{out->des_pos, out->des_quat, out->des_linvel, out->des_angvel, out->joint_command,
                      out->foot_position, out->foot_velocity, out->foot_acceleration, out->surface_normal, out->phase,
                      out->support_leg, out->leg_mode};
  const uint8_t *d_msg = messages;
  const int64_t *d_off = offsets;
  int32_t *d_st = status;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    for (size_t k = 0; k < B; k++)
      if (offsets[k + 1] < offsets[k] || offsets[0] < 0) return QLAMD_ERR_INVALID_ARGUMENT;
    const size_t nbytes = (size_t)(offsets[B] - offsets[0]);
    // inputs first, outputs after them: a small call is then one copy each way over a tight span
    const int i_off = sg.add(offsets, (B + 1) * 8, true, false);
    const int i_msg = sg.add(messages + offsets[0], nbytes, true, false);
    int i_d[kD];
    for (int k = 0; k < kD; k++) i_d[k] = sg.add(hostd[k], B * 8 * (size_t)width[k], false, true);
    const int i_sup = sg.add(out->support_leg, B * 4, false, true);
    const int i_mode = sg.add(out->leg_mode, B * 4, false, true);
    const int i_st = sg.add(status, B * 4, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    o = RobotStateOutPtrs{sg.dev<double>(i_d[0]), sg.dev<double>(i_d[1]), sg.dev<double>(i_d[2]), sg.dev<double>(i_d[3]),
                          sg.dev<double>(i_d[4]), sg.dev<double>(i_d[5]), sg.dev<double>(i_d[6]), sg.dev<double>(i_d[7]),
                          sg.dev<double>(i_d[8]), sg.dev<double>(i_d[9]), sg.dev<uint8_t>(i_sup), sg.dev<uint8_t>(i_mode)};
    d_st = sg.dev<int32_t>(i_st);
    d_off = sg.dev<const int64_t>(i_off);
    d_msg = (const uint8_t *)(sg.base + sg.items[i_msg].off) - offsets[0]; // the kernel indexes with the caller's offsets
  }
  if (!ctx->wire_tpl) { // zero = "no template yet": the first launch walks every message
    if (hipMalloc((void **)&ctx->wire_tpl, 2 * kTplWords * sizeof(uint32_t)) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
    if (hipMemsetAsync(ctx->wire_tpl, 0, 2 * kTplWords * sizeof(uint32_t), st) != hipSuccess) return QLAMD_ERR_HIP;
  }
  const uint32_t *tpl_in = ctx->wire_tpl + kTplWords * ctx->wire_flip;
  uint32_t *tpl_out = ctx->wire_tpl + kTplWords * (ctx->wire_flip ^ 1);
  ctx->wire_flip ^= 1;
  hipLaunchKernelGGL(robot_state_unpack_kernel, dim3((unsigned)((batch + kWireMsgsPerBlock - 1) / kWireMsgsPerBlock)),
                     dim3(64), kWireLdsBytes, st, d_msg, d_off, batch, o, d_st, tpl_in, tpl_out);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

void qlamd_ik_default_params(qlamd_ik_params *p) {
  if (!p) return;
  p->d = 0.1; p->l1 = 0.25; p->l2 = 0.25;       // quadrupedkinematics.cpp:383-385
  // setLimbConfigure("><"), quadruped_state.cpp:61,385-390: LF IN_LEFT, RF OUT_LEFT, RH IN_LEFT, LH OUT_LEFT
  p->limb_config[0] = QLAMD_IK_IN_LEFT; p->limb_config[1] = QLAMD_IK_OUT_LEFT;
  p->limb_config[2] = QLAMD_IK_IN_LEFT; p->limb_config[3] = QLAMD_IK_OUT_LEFT;
}

int qlamd_leg_inverse_kinematics_batch(qlamd_context *ctx, const qlamd_ik_params *params, const double *foot_position,
                                       const double *joint_position_last, int64_t batch, double *joint_position,
                                       uint8_t *ok, int memory, void *stream) {
  if (!ctx || !foot_position || !joint_position || batch < 0) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!params) return QLAMD_ERR_NOT_LOADED;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  IkGeom G;
  G.g[0] = params->d; G.g[1] = params->l1; G.g[2] = params->l2;
  for (int l = 0; l < 4; l++) {
    if (params->limb_config[l] > 3) return QLAMD_ERR_INVALID_ARGUMENT;
    G.config[l] = params->limb_config[l];
  }
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)batch;
  const double *d_foot = foot_position, *d_last = joint_position_last;
  double *d_q = joint_position;
  uint8_t *d_ok = ok;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    const int a = sg.add(foot_position, B * 96, true, false), b2 = sg.add(joint_position_last, B * 96, true, false);
    const int c = sg.add(joint_position, B * 96, false, true), d = sg.add(ok, B * 4, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    d_foot = sg.dev<const double>(a); d_last = sg.dev<const double>(b2); d_q = sg.dev<double>(c); d_ok = sg.dev<uint8_t>(d);
  }
  hipLaunchKernelGGL(leg_ik_kernel, dim3((unsigned)((4 * batch + 63) / 64)), dim3(64), 0, st, ctx->d_params, G, d_foot,
                     d_last, batch, d_q, d_ok);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  return memory == QLAMD_MEM_HOST ? sg.finish(st) : QLAMD_OK;
}

int qlamd_qp_solve_batch(qlamd_context *ctx, int n, int p, int m, const double *G, const double *g0,
                         const double *CE, const double *ce0, const double *CI, const double *ci0,
                         int64_t batch, double *x, double *objective, int32_t *status, int memory,
                         void *stream) {
  if (!ctx || batch < 0 || !G || !g0 || !x || !status) return QLAMD_ERR_INVALID_ARGUMENT;
  if (n < 1 || n > 12 || p < 0 || p > 2 || m < 0 || m > 48) return QLAMD_ERR_INVALID_ARGUMENT;
  if (m > 24 && (p > 1 || getenv("QLAMD_QP_ONE_LANE"))) return QLAMD_ERR_INVALID_ARGUMENT; // 25..48 rows: cooperative kernel only
  if ((p > 0 && (!CE || !ce0)) || (m > 0 && (!CI || !ci0))) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
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
  if (p <= 1 && !getenv("QLAMD_QP_ONE_LANE")) {
    // lane-cooperative kernel (at most one equality column: what every caller in the reference passes)
    const unsigned cgrid = (unsigned)((batch + coop::kQpCoopRows - 1) / coop::kQpCoopRows);
    auto launch = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(cgrid), dim3(64), 0, st, n, p, m, dG, dg0, dCE, dce0, dCI, dci0, batch, dx, dobj, dst);
    };
    if (m > 24) {
      if (n <= 6) launch(qp_coop_kernel<6, 3>); else launch(qp_coop_kernel<12, 3>);
    } else {
      if (n <= 6) launch(qp_coop_kernel<6, 2>); else launch(qp_coop_kernel<12, 2>);
    }
  } else {
    // one lane per problem, following solve_quadprog step by step (two equality columns, or on request)
    const size_t lds = (size_t)kQpPerWave * QpGi::kTotal * sizeof(double);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void *)qp_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return QLAMD_ERR_HIP;
    const unsigned grid = (unsigned)((batch + kQpPerWave - 1) / kQpPerWave);
    hipLaunchKernelGGL(qp_solve_kernel, dim3(grid), dim3(64), lds, st, n, p, m, dG, dg0, dCE, dce0, dCI, dci0, batch, dx,
                       dobj, dst);
  }
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  if (memory == QLAMD_MEM_HOST) {
    if (hipMemcpyAsync(x, dx, B * n * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (objective && hipMemcpyAsync(objective, dobj, B * 8, hipMemcpyDeviceToHost, st) != hipSuccess)
      return QLAMD_ERR_HIP;
    if (hipMemcpyAsync(status, dst, B * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return QLAMD_ERR_HIP;
  }
  return QLAMD_OK;
}

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
  const bool fused = getenv("QLAMD_WB_SPLIT") == nullptr;
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

int qlamd_full_tick_batch(qlamd_context *ctx, const qlamd_swing_params *swing, const qlamd_joint_pid_params *pid,
                          const qlamd_tick_batch *io, double period, int index_quirk, int64_t batch, int memory,
                          void *stream) {
  if (!ctx || !io || batch < 0) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!swing || !pid) return QLAMD_ERR_NOT_LOADED;
  if (!io->messages || !io->offsets || !io->joint_position || !io->joint_velocity || !io->joint_velocity_oldest ||
      !io->base_position || !io->base_orientation || !io->base_linear_velocity || !io->base_angular_velocity ||
      !io->contact || !io->limb_state || !io->store_flag || !io->stored_joint_position || !io->leg_mode || !io->support ||
      !io->pid_error_last || !io->pid_error_integral || !io->joint_effort || !io->status || !io->message_status)
    return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)batch;
  qlamd_tick_batch d = *io;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    for (size_t k = 0; k < B; k++)
      if (io->offsets[k + 1] < io->offsets[k] || io->offsets[0] < 0) return QLAMD_ERR_INVALID_ARGUMENT;
    const size_t nbytes = (size_t)(io->offsets[B] - io->offsets[0]);
    const int i_off = sg.add(io->offsets, (B + 1) * 8, true, false);
    const int i_msg = sg.add(io->messages + io->offsets[0], nbytes ? nbytes : 1, true, false);
    const int i_in[8] = {sg.add(io->joint_position, B * 96, true, false), sg.add(io->joint_velocity, B * 96, true, false),
                         sg.add(io->joint_velocity_oldest, B * 96, true, false), sg.add(io->base_position, B * 24, true, false),
                         sg.add(io->base_orientation, B * 32, true, false), sg.add(io->base_linear_velocity, B * 24, true, false),
                         sg.add(io->base_angular_velocity, B * 24, true, false), sg.add(io->contact, B * 4, true, false)};
    const int i_io[7] = {sg.add(io->limb_state, B * 4, true, true), sg.add(io->store_flag, B * 4, true, true),
                         sg.add(io->stored_joint_position, B * 96, true, true), sg.add(io->leg_mode, B * 4, true, true),
                         sg.add(io->pid_error_last, B * 96, true, true), sg.add(io->pid_error_integral, B * 96, true, true),
                         sg.add(io->support, B * 4, true, true)};
    const int i_out[4] = {sg.add(io->joint_effort, B * 96, false, true), sg.add(io->leg_state_code, B * 4, false, true),
                          sg.add(io->status, B * 4, false, true), sg.add(io->message_status, B * 4, false, true)};
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    d.offsets = sg.dev<const int64_t>(i_off);
    d.messages = (const uint8_t *)(sg.base + sg.items[i_msg].off) - io->offsets[0];
    d.joint_position = sg.dev<const double>(i_in[0]); d.joint_velocity = sg.dev<const double>(i_in[1]);
    d.joint_velocity_oldest = sg.dev<const double>(i_in[2]); d.base_position = sg.dev<const double>(i_in[3]);
    d.base_orientation = sg.dev<const double>(i_in[4]); d.base_linear_velocity = sg.dev<const double>(i_in[5]);
    d.base_angular_velocity = sg.dev<const double>(i_in[6]); d.contact = sg.dev<const uint8_t>(i_in[7]);
    d.limb_state = sg.dev<int8_t>(i_io[0]); d.store_flag = sg.dev<uint8_t>(i_io[1]);
    d.stored_joint_position = sg.dev<double>(i_io[2]); d.leg_mode = sg.dev<uint8_t>(i_io[3]);
    d.pid_error_last = sg.dev<double>(i_io[4]); d.pid_error_integral = sg.dev<double>(i_io[5]);
    d.support = sg.dev<uint8_t>(i_io[6]);
    d.joint_effort = sg.dev<double>(i_out[0]); d.leg_state_code = sg.dev<int8_t>(i_out[1]);
    d.status = sg.dev<int32_t>(i_out[2]); d.message_status = sg.dev<int32_t>(i_out[3]);
  }
  // intermediates: what the message delivers and what the state machine decides
  enum { kPos, kQuat, kLin, kAng, kCmd, kFootP, kFootV, kPhase, kMsgSup, kMsgMode, kFootstep, kCode, kN };
  const size_t sz[kN] = {B * 24, B * 32, B * 24, B * 24, B * 96, B * 96, B * 96, B * 32, B * 4, B * 4, B * 4, B * 4};
  size_t off[kN], total = 0;
  for (int k = 0; k < kN; k++) { off[k] = total; total += align256(sz[k]); }
  if (ctx->tick_ws_bytes < total) {
    if (ctx->tick_ws) (void)hipFree(ctx->tick_ws);
    ctx->tick_ws = nullptr; ctx->tick_ws_bytes = 0;
    if (hipMalloc(&ctx->tick_ws, total) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
    ctx->tick_ws_bytes = total;
  }
  char *w = (char *)ctx->tick_ws;
  const auto D = [&](int k) { return (double *)(w + off[k]); };
  const auto U = [&](int k) { return (uint8_t *)(w + off[k]); };
  int rc;
  // 1. baseCommandCallback: message -> desired state, targets, leg modes
  qlamd_robot_state_fields f{};
  f.des_pos = D(kPos); f.des_quat = D(kQuat); f.des_linvel = D(kLin); f.des_angvel = D(kAng);
  f.joint_command = D(kCmd); f.foot_position = D(kFootP); f.foot_velocity = D(kFootV); f.phase = D(kPhase);
  f.support_leg = U(kMsgSup); f.leg_mode = U(kMsgMode);
  rc = qlamd_robot_state_unpack_batch(ctx, d.messages, d.offsets, batch, &f, d.message_status, QLAMD_MEM_DEVICE, stream);
  if (rc != QLAMD_OK) return rc;
  hipLaunchKernelGGL(tick_leg_mode_kernel, dim3((unsigned)((4 * batch + 255) / 256)), dim3(256), 0, st, U(kMsgMode), d.leg_mode,
                     U(kFootstep), 4 * batch);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  // 2. footContactsCallback + the switch of update(): support legs, held joint commands, nudged foot targets
  qlamd_leg_state_batch ls{U(kMsgSup), D(kPhase), U(kFootstep), d.contact, d.joint_position, d.limb_state, d.store_flag,
                           d.stored_joint_position, D(kCmd), D(kFootP), d.support,
                           d.leg_state_code ? d.leg_state_code : (int8_t *)(w + off[kCode])};
  rc = qlamd_leg_state_machine_batch(ctx, &ls, index_quirk, batch, QLAMD_MEM_DEVICE, stream);
  if (rc != QLAMD_OK) return rc;
  // 3. balance solve for the support legs (all 12 efforts written: 0 for the others)
  qlamd_state_batch sb{d.joint_position, d.base_position, d.base_orientation, d.base_linear_velocity, d.base_angular_velocity,
                       D(kPos), D(kQuat), D(kLin), D(kAng), d.support, nullptr};
  rc = qlamd_balance_solve_batch(ctx, &sb, batch, d.joint_effort, nullptr, d.status, QLAMD_MEM_DEVICE, stream);
  if (rc != QLAMD_OK) return rc;
  // 4. swing branch for the legs that do not support
  qlamd_swing_params sp = *swing;
  sp.period = period;
  const qlamd_swing_batch sw{d.joint_position, d.joint_velocity, d.joint_velocity_oldest, D(kFootP), D(kFootV), d.support, nullptr};
  const qlamd_swing_branch_extra ex{d.base_orientation, D(kCmd), d.leg_mode, d.pid_error_last, d.pid_error_integral};
  rc = qlamd_swing_branch_batch(ctx, &sp, pid, &sw, &ex, period, batch, d.joint_effort, QLAMD_MEM_DEVICE, stream);
  if (rc != QLAMD_OK) return rc;
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

#ifdef QLAMD_STAMPS
int qlamd_debug_stamps(unsigned long long *out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(qlamd::coop::g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

const char *qlamd_strerror(int code) {
  switch (code) {
    case QLAMD_OK: return "ok";
    case QLAMD_ERR_INVALID_ARGUMENT: return "invalid argument";
    case QLAMD_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case QLAMD_ERR_HIP: return "HIP runtime error";
    case QLAMD_ERR_NOT_LOADED: return "parameters not loaded";
    case QLAMD_ERR_OUT_OF_MEMORY: return "out of device memory";
    default: return "unknown error";
  }
}

int qlamd_context_create(const qlamd_balance_params *params, const qlamd_robot_model *model, int device,
                         qlamd_context **out) {
  if (!out) return QLAMD_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!params) return QLAMD_ERR_NOT_LOADED;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return QLAMD_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return QLAMD_ERR_INVALID_ARGUMENT;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return QLAMD_ERR_HIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return QLAMD_ERR_NO_DEVICE;
  qlamd_context *ctx = new (std::nothrow) qlamd_context;
  if (!ctx) return QLAMD_ERR_OUT_OF_MEMORY;
  ctx->device = device;
  ctx->rpw_override = 0;
  ctx->num_cu = prop.multiProcessorCount;
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  ctx->pinned = nullptr;
  ctx->pinned_bytes = 0;
  ctx->wire_tpl = nullptr;
  ctx->wire_flip = 0;
  ctx->tick_ws = nullptr;
  ctx->tick_ws_bytes = 0;
  qlamd_robot_model m;
  if (model) m = *model; else default_robot_model(&m);
  build_device_params(*params, m, &ctx->params);
  {
    // base_link inertia moved from its centre of mass to the base origin: I + m (|c|^2 1 - c c')
    const double bm = m.base_mass, *c = m.base_com, *I = m.base_inertia, cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    ctx->base_m = bm;
    for (int a = 0; a < 3; a++) ctx->base_h[a] = bm * c[a];
    ctx->base_I[0] = I[0] + bm * (cc - c[0] * c[0]); ctx->base_I[1] = I[1] - bm * c[0] * c[1]; ctx->base_I[2] = I[2] - bm * c[0] * c[2];
    ctx->base_I[3] = I[3] + bm * (cc - c[1] * c[1]); ctx->base_I[4] = I[4] - bm * c[1] * c[2];
    ctx->base_I[5] = I[5] + bm * (cc - c[2] * c[2]);
  }
  ctx->d_params = nullptr;
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&ctx->d_params, sizeof(DeviceParams)) != hipSuccess ||
      hipMemcpy(ctx->d_params, &ctx->params, sizeof(DeviceParams), hipMemcpyHostToDevice) != hipSuccess) {
    if (ctx->d_params) (void)hipFree(ctx->d_params);
    delete ctx;
    return QLAMD_ERR_HIP;
  }
  *out = ctx;
  return QLAMD_OK;
}

void qlamd_context_destroy(qlamd_context *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->wire_tpl) (void)hipFree(ctx->wire_tpl);
  if (ctx->tick_ws) (void)hipFree(ctx->tick_ws);
  if (ctx->d_params) (void)hipFree(ctx->d_params);
  delete ctx;
}

int qlamd_set_robots_per_wave(qlamd_context *ctx, int rpw) {
  if (!ctx || !(rpw == 0 || rpw == 4 || rpw == 16 || rpw == 64)) return QLAMD_ERR_INVALID_ARGUMENT;
  ctx->rpw_override = rpw;
  return QLAMD_OK;
}

static int balance_impl(qlamd_context *ctx, const qlamd_state_batch *in_user, const double *wrench, int64_t batch,
                        double *joint_effort, double *contact_force, int32_t *status, int memory, void *stream) {
  if (!ctx || !in_user || batch < 0 || !joint_effort || !status) return QLAMD_ERR_INVALID_ARGUMENT;
  qlamd_state_batch filled = *in_user;
  const qlamd_state_batch *in = &filled;
  if (!filled.joint_position || !filled.base_orientation || !filled.support_leg) return QLAMD_ERR_INVALID_ARGUMENT;
  if (wrench) {
    // force distribution only: the base pose / twist fields are not read for arithmetic; point
    // them at valid memory of sufficient size (joint_position is [B][12])
    filled.base_position = filled.base_linear_velocity = filled.base_angular_velocity = filled.joint_position;
    filled.desired_position = filled.desired_linear_velocity = filled.desired_angular_velocity = filled.joint_position;
    filled.desired_orientation = filled.base_orientation;
  } else if (!filled.base_position || !filled.base_linear_velocity || !filled.base_angular_velocity ||
             !filled.desired_position || !filled.desired_orientation || !filled.desired_linear_velocity ||
             !filled.desired_angular_velocity) {
    return QLAMD_ERR_INVALID_ARGUMENT;
  }
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)batch;

  StatePtrs s;
  double *d_tau = joint_effort, *d_grf = contact_force;
  int32_t *d_status = status;
  bool small_host = false;
  size_t out_off = 0, out_bytes = 0;
  if (memory == QLAMD_MEM_HOST) {
    // one staging slab: inputs then outputs, 256-byte aligned pieces
    // (with an external wrench the pose / twist fields alias the head of joint_position)
    const size_t rec = 24, qrec = 32;
    const size_t sz[12] = {B * 96, B * rec, B * qrec, B * rec, B * rec, B * rec, B * qrec, B * rec, B * rec, B * 4,
                           in->surface_normal ? B * 96 : 0, wrench ? B * 48 : 0};
    const void *src[12] = {in->joint_position, in->base_position, in->base_orientation,
                           in->base_linear_velocity, in->base_angular_velocity, in->desired_position,
                           in->desired_orientation, in->desired_linear_velocity,
                           in->desired_angular_velocity, in->support_leg, in->surface_normal, wrench};
    size_t off[15], total = 0;
    for (int k = 0; k < 12; k++) { off[k] = total; total += align256(sz[k]); }
    off[12] = total; total += align256(B * 96);
    off[13] = total; total += align256(B * 96);
    off[14] = total; total += align256(B * 4);
    int rc = ensure_ws(ctx, total);
    if (rc != QLAMD_OK) return rc;
    char *w = (char *)ctx->ws;
    small_host = total <= kSmallHostCall && ensure_pinned(ctx, kSmallHostCall) == QLAMD_OK;
    out_off = off[12];
    out_bytes = total - off[12];
    if (small_host) {
      // a single robot or a few: a dozen separate pageable copies cost ~10 us each; pack, copy once
      char *h = (char *)ctx->pinned;
      for (int k = 0; k < 12; k++)
        if (sz[k]) memcpy(h + off[k], src[k], sz[k]);
      if (hipMemcpyAsync(w, h, off[12], hipMemcpyHostToDevice, st) != hipSuccess) return QLAMD_ERR_HIP;
    } else {
      for (int k = 0; k < 12; k++)
        if (sz[k] && hipMemcpyAsync(w + off[k], src[k], sz[k], hipMemcpyHostToDevice, st) != hipSuccess)
          return QLAMD_ERR_HIP;
    }
    s = StatePtrs{(const double *)(w + off[0]), (const double *)(w + off[1]), (const double *)(w + off[2]),
                  (const double *)(w + off[3]), (const double *)(w + off[4]), (const double *)(w + off[5]),
                  (const double *)(w + off[6]), (const double *)(w + off[7]), (const double *)(w + off[8]),
                  (const uint8_t *)(w + off[9]), in->surface_normal ? (const double *)(w + off[10]) : nullptr,
                  wrench ? (const double *)(w + off[11]) : nullptr};
    d_tau = (double *)(w + off[12]);
    d_grf = contact_force ? (double *)(w + off[13]) : nullptr;
    d_status = (int32_t *)(w + off[14]);
  } else {
    s = StatePtrs{in->joint_position, in->base_position, in->base_orientation, in->base_linear_velocity,
                  in->base_angular_velocity, in->desired_position, in->desired_orientation,
                  in->desired_linear_velocity, in->desired_angular_velocity, in->support_leg,
                  in->surface_normal, wrench};
  }

  hipError_t e;
  switch (pick_rpw(ctx, batch)) {
    case 4: {
      const unsigned grid = (unsigned)((batch + 3) / 4);
      if (s.normals)
        hipLaunchKernelGGL(balance_coop_kernel<true>, dim3(grid), dim3(64), 0, st, ctx->d_params, s, batch, d_tau,
                           d_grf, d_status);
      else
        hipLaunchKernelGGL(balance_coop_kernel<false>, dim3(grid), dim3(64), 0, st, ctx->d_params, s, batch, d_tau,
                           d_grf, d_status);
      e = hipGetLastError();
      break;
    }
    case 16: e = launch_balance<16>(ctx, s, batch, d_tau, d_grf, d_status, st); break;
    default: e = launch_balance<64>(ctx, s, batch, d_tau, d_grf, d_status, st); break;
  }
  if (e != hipSuccess) return QLAMD_ERR_HIP;

  if (memory == QLAMD_MEM_HOST && small_host) {
    char *h = (char *)ctx->pinned, *w = (char *)ctx->ws;
    if (hipMemcpyAsync(h + out_off, w + out_off, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return QLAMD_ERR_HIP;
    memcpy(joint_effort, h + ((char *)d_tau - w), B * 96);
    if (contact_force) memcpy(contact_force, h + ((char *)d_grf - w), B * 96);
    memcpy(status, h + ((char *)d_status - w), B * 4);
  } else if (memory == QLAMD_MEM_HOST) {
    if (hipMemcpyAsync(joint_effort, d_tau, B * 96, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (contact_force &&
        hipMemcpyAsync(contact_force, d_grf, B * 96, hipMemcpyDeviceToHost, st) != hipSuccess)
      return QLAMD_ERR_HIP;
    if (hipMemcpyAsync(status, d_status, B * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return QLAMD_ERR_HIP;
  }
  return QLAMD_OK;
}

int qlamd_balance_solve_batch(qlamd_context *ctx, const qlamd_state_batch *in, int64_t batch, double *joint_effort,
                              double *contact_force, int32_t *status, int memory, void *stream) {
  return balance_impl(ctx, in, nullptr, batch, joint_effort, contact_force, status, memory, stream);
}

int qlamd_force_distribution_batch(qlamd_context *ctx, const double *joint_position, const double *base_orientation,
                                   const uint8_t *support_leg, const double *surface_normal,
                                   const double *virtual_wrench, int64_t batch, double *joint_effort,
                                   double *contact_force, int32_t *status, int memory, void *stream) {
  if (!virtual_wrench) return QLAMD_ERR_INVALID_ARGUMENT;
  qlamd_state_batch in;
  memset(&in, 0, sizeof(in));
  in.joint_position = joint_position;
  in.base_orientation = base_orientation;
  in.support_leg = support_leg;
  in.surface_normal = surface_normal;
  return balance_impl(ctx, &in, virtual_wrench, batch, joint_effort, contact_force, status, memory, stream);
}

int qlamd_virtual_wrench_batch(qlamd_context *ctx, const qlamd_state_batch *in, int64_t batch, double *wrench,
                               int memory, void *stream) {
  if (!ctx || !in || batch < 0 || !wrench) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!in->base_position || !in->base_orientation || !in->base_linear_velocity || !in->base_angular_velocity ||
      !in->desired_position || !in->desired_orientation || !in->desired_linear_velocity || !in->desired_angular_velocity)
    return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)batch;
  StatePtrs s{in->joint_position, in->base_position, in->base_orientation, in->base_linear_velocity,
              in->base_angular_velocity, in->desired_position, in->desired_orientation,
              in->desired_linear_velocity, in->desired_angular_velocity, in->support_leg, nullptr, nullptr};
  double *d_w = wrench;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    const int a1 = sg.add(in->base_position, B * 24, true, false), a2 = sg.add(in->base_orientation, B * 32, true, false);
    const int a3 = sg.add(in->base_linear_velocity, B * 24, true, false), a4 = sg.add(in->base_angular_velocity, B * 24, true, false);
    const int a5 = sg.add(in->desired_position, B * 24, true, false), a6 = sg.add(in->desired_orientation, B * 32, true, false);
    const int a7 = sg.add(in->desired_linear_velocity, B * 24, true, false);
    const int a8 = sg.add(in->desired_angular_velocity, B * 24, true, false), o = sg.add(wrench, B * 48, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    s = StatePtrs{nullptr, sg.dev<const double>(a1), sg.dev<const double>(a2), sg.dev<const double>(a3),
                  sg.dev<const double>(a4), sg.dev<const double>(a5), sg.dev<const double>(a6), sg.dev<const double>(a7),
                  sg.dev<const double>(a8), nullptr, nullptr, nullptr};
    d_w = sg.dev<double>(o);
  }
  const unsigned grid = (unsigned)((batch + 63) / 64);
  hipLaunchKernelGGL(virtual_wrench_kernel, dim3(grid), dim3(64), 0, st, ctx->d_params, s, batch, d_w);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  return memory == QLAMD_MEM_HOST ? sg.finish(st) : QLAMD_OK;
}

int qlamd_leg_kinematics_batch(qlamd_context *ctx, const double *joint_position, const double *base_orientation,
                               int64_t batch, double *foot_position, double *jacobian, double *gravity_torque,
                               int memory, void *stream) {
  if (!ctx || !joint_position || !base_orientation || batch < 0) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)batch;
  const double *d_q = joint_position, *d_quat = base_orientation;
  double *d_f = foot_position, *d_j = jacobian, *d_g = gravity_torque;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    const int a1 = sg.add(joint_position, B * 96, true, false), a2 = sg.add(base_orientation, B * 32, true, false);
    const int o1 = sg.add(foot_position, B * 96, false, true), o2 = sg.add(jacobian, B * 288, false, true);
    const int o3 = sg.add(gravity_torque, B * 96, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    d_q = sg.dev<const double>(a1); d_quat = sg.dev<const double>(a2);
    d_f = sg.dev<double>(o1); d_j = sg.dev<double>(o2); d_g = sg.dev<double>(o3);
  }
  const unsigned grid = (unsigned)((4 * batch + 63) / 64);
  hipLaunchKernelGGL(leg_kinematics_kernel, dim3(grid), dim3(64), 0, st, ctx->d_params, d_q, d_quat, batch, d_f, d_j, d_g);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  return memory == QLAMD_MEM_HOST ? sg.finish(st) : QLAMD_OK;
}

} // extern "C"
