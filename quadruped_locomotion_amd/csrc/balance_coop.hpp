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

#include "force_qp_coop.hpp"

namespace qlamd {
namespace coop {

struct CoopPtrs {
  const double *q, *pos, *quat, *linvel, *angvel, *dpos, *dquat, *dlinvel, *dangvel;
  const uint8_t *stance;
  const double *normals;
  const double *wrench; // [B][6] or NULL: externally supplied (F_B, T_B)
  const uint8_t *live;  // [B] or NULL: 0 = leave this robot alone (whole tick: no command in force), nothing is written
  int support_only;     // whole tick: write the efforts of the support legs only (the swing branch owns the others)
  int32_t *iterations;  // [B] or NULL: outer iterations of each robot's QP (a placement hint for the next control step)
  // kWarm only: the working set each robot's QP starts from (its final working set of the previous control step) and
  // where this step's final working set goes; bit 5 leg + kind
  const uint32_t *prev_working_set;
  uint32_t *working_set;
  uint32_t *warm_retries; // the context's count of rejected warm starts (QLAMD_COUNTER_WARM_RETRIES), or NULL
  int record_doubles;     // 0: one array per field; R: the double fields are records of R doubles (QLAMD_STATE_RECORDS)
};

// Support legs first.  The QP's variables are the contact forces of the legs that support; with the legs of a robot laid
// out in the row in the order [support legs ascending, then the others] a robot on two legs (a trot) is a 6-variable problem
// in rows 0 and 1 -- force_qp_coop<..., kLegs = 2>: every broadcast product, every rank-one update and the inversion of G
// half as long -- and a robot on four legs is laid out as always.  slot_legs(m): the leg behind each of the four slots for
// the support mask m, two bits per slot (0xE4 = identity).
__host__ __device__ constexpr unsigned slot_legs_of(unsigned m) {
  unsigned perm = 0, k = 0;
  for (unsigned l = 0; l < 4; l++)
    if ((m >> l) & 1u) perm |= l << (2 * k++);
  for (unsigned l = 0; l < 4; l++)
    if (!((m >> l) & 1u)) perm |= l << (2 * k++);
  return perm;
}
__host__ __device__ constexpr unsigned long long slot_legs_table(int half) {
  unsigned long long t = 0;
  for (int m = 0; m < 8; m++) t |= (unsigned long long)slot_legs_of(8 * half + m) << (8 * m);
  return t;
}
__device__ __forceinline__ unsigned slot_legs(unsigned m) {
  constexpr unsigned long long lo = slot_legs_table(0), hi = slot_legs_table(1);
  return (unsigned)(((m & 8u) ? hi : lo) >> (8 * (m & 7u))) & 0xFFu;
}
// a working set (kKinds bits per leg) between the caller's leg order and the row's slot order
template <int kKinds, class mask_t>
__device__ __forceinline__ mask_t working_set_to_slots(mask_t by_leg, unsigned perm) {
  constexpr mask_t rows = ((mask_t)1 << kKinds) - 1;
  mask_t out = 0;
#pragma unroll
  for (int sl = 0; sl < 4; sl++) out |= ((by_leg >> (kKinds * ((perm >> (2 * sl)) & 3u))) & rows) << (kKinds * sl);
  return out;
}
template <int kKinds, class mask_t>
__device__ __forceinline__ mask_t working_set_to_legs(mask_t by_slot, unsigned perm) {
  constexpr mask_t rows = ((mask_t)1 << kKinds) - 1;
  mask_t out = 0;
#pragma unroll
  for (int sl = 0; sl < 4; sl++) out |= ((by_slot >> (kKinds * sl)) & rows) << (kKinds * ((perm >> (2 * sl)) & 3u));
  return out;
}
// my value of a per-lane quantity laid out by leg, fetched from the lane of the leg behind my slot (same component)
__device__ __forceinline__ double from_leg_lane(double v, int aleg) {
  const int addr = (((int)threadIdx.x & 48) | (4 * aleg) | ((int)threadIdx.x & 3)) << 2;
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)),
                          __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}

// One robot per 16-lane row.  lds_tab: 256-double model table; lds_row: this robot's private
// LDS block of kCoopLdsDoubles doubles (N* export for the refinement, row export for drops); lds_nrm: the wavefront's
// table of constraint normals, kCoopNrmDoubles doubles ([row kind][lane]).
constexpr int kCoopNrmDoubles = 12 * 64; // 5 row kinds + parked Jacobian row (3), gravity torque (3), output index and slot order

// kParkInputs (the 168-register form): what only the assembly of G and g0 reads -- the foot position and the wrench -- waits in
// LDS instead of in registers while the first of the two forms of the QP runs (see `solve` below)
// Returns whether this row's warm start was rejected (kWarm: the answer reached from the set handed in failed the final check of
// force_qp_coop) -- coop_robot_checked() below then solves the robot again with `cold` set: the set handed in is ignored.
// kThroughput (the 168-register form of the kernels): a warm start installs its rows one after the other, not by rounds
// kSmallForm: the kernel carries the 6-variable form of the QP for wavefronts whose four robots stand on at most two legs (the
// placed entries, whose sorted placement groups robots by class); without it (the plain entry: who shares a wavefront is an
// accident of the batch order there, and the second form cost it 3-4 % on batches of robots on four legs) every wavefront
// takes the 12-variable form, which gives a robot on two legs the same result bit for bit (its padding rows add exact zeros)
template <bool kPerLeg, int kBlock = 64, bool kWarm = false, bool kParkInputs = false, bool kThroughput = false, bool kSmallForm = true>
__device__ __forceinline__ bool coop_robot(const DeviceParams &P, const CoopPtrs &s, int64_t irobot, bool robot_live_in,
                                           double *lds_tab, double *lds_row, double *lds_nrm,
                                           double *__restrict__ tau_out,
                                           double *__restrict__ grf_out, int32_t *__restrict__ status_out, bool cold = false) {
  bool robot_live = robot_live_in;
  const int lr = threadIdx.x & 15;   // lane in row
  const int leg = lr >> 2, c = lr & 3; // leg: my SLOT in the row (the leg behind it: aleg, below)
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
  const int64_t i_in = irobot; // (below the QP the index comes back from LDS as `i`: one value less across the loop)
  // Strides of the nine double fields, in doubles: their widths (12 / 4 / 3: one array per field, the layout of
  // hardware_interface::RobotStateHandle::Data) or, with QLAMD_OPT_STATE_LAYOUT = QLAMD_STATE_RECORDS, one record length for all
  // (every field pointer addresses the same [B][R] buffer at its field's offset): a permuted launch then touches three lines
  // per robot instead of nine partially used ones (s.record_doubles, 0 = per-field arrays).  The support flags stay an array of
  // their own (the shadow wavefronts read them too).
  const int64_t kS12 = s.record_doubles ? s.record_doubles : 12, kS4 = s.record_doubles ? s.record_doubles : 4,
                kS3 = s.record_doubles ? s.record_doubles : 3, kSB = 4;
  double quat[4], dquat[4], pos[3], linvel[3], angvel[3], dpos[3], dlinvel[3], dangvel[3];
  {
    const double2 *a2 = reinterpret_cast<const double2 *>(s.quat + kS4 * i_in);
    const double2 *b2 = reinterpret_cast<const double2 *>(s.dquat + kS4 * i_in);
    double2 v = a2[0]; quat[0] = v.x; quat[1] = v.y;
    v = a2[1]; quat[2] = v.x; quat[3] = v.y;
    v = b2[0]; dquat[0] = v.x; dquat[1] = v.y;
    v = b2[1]; dquat[2] = v.x; dquat[3] = v.y;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      pos[k] = s.pos[kS3 * i_in + k]; linvel[k] = s.linvel[kS3 * i_in + k]; angvel[k] = s.angvel[kS3 * i_in + k];
      dpos[k] = s.dpos[kS3 * i_in + k]; dlinvel[k] = s.dlinvel[kS3 * i_in + k]; dangvel[k] = s.dangvel[kS3 * i_in + k];
    }
  }
  const uint32_t sm = *reinterpret_cast<const uint32_t *>(s.stance + kSB * i_in);
  double qj = s.q[kS12 * i_in + (comp ? myidx : 0)];
  const uint8_t alive = s.live ? s.live[i_in] : (uint8_t)1;
  unsigned warm_set = 0u;
  bool build_set = false;
  if constexpr (kWarm) warm_set = (s.prev_working_set && !cold) ? s.prev_working_set[i_in] : 0u;
  double wr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}; // externally supplied (F_B, T_B), if any: issued with the rest
  if (s.wrench) {
#pragma unroll
    for (int k = 0; k < 6; k++) wr[k] = s.wrench[6 * i_in + k];
  }
  double nWl[3] = {0.0, 0.0, 1.0}; // caller-supplied surface normal of my leg (world frame), if any
  if (kPerLeg) { nWl[0] = s.normals[12 * i_in + 3 * leg]; nWl[1] = s.normals[12 * i_in + 3 * leg + 1]; nWl[2] = s.normals[12 * i_in + 3 * leg + 2]; }
  // (all loads above are in flight before the first of them is consumed)
  robot_live = robot_live && alive != 0;
  const unsigned stance = robot_live ? (((sm & 0xFFu) ? 1u : 0u) | ((sm & 0xFF00u) ? 2u : 0u) |
                                        ((sm & 0xFF0000u) ? 4u : 0u) | ((sm & 0xFF000000u) ? 8u : 0u))
                                     : 0u;
  const int nS = __popc(stance);
  if constexpr (kWarm) {
    // A working set remembers the support legs it was reached with (bits 20..23; 0: not recorded).  A robot that has changed
    // them since -- a trot entering or leaving double support -- starts cold: on the bench's trot trajectory such a robot
    // needs 11.2 installs / drops / passes from its stale set against 9.5 passes from the empty one (2 -> 4 legs), 6.0 against
    // 3.4 (4 -> 2): profiles/r6/trajectory_stats.txt
    const unsigned from = (warm_set >> 20) & 0xFu;
    warm_set = (from != 0u && from != stance) ? 0u : (warm_set & 0xFFFFFu);
    // ... and BUILDS a set by rounds instead of adding a row a pass (force_qp_coop.hpp, kGreedy), as does a robot on its first
    // step (a word of zeros: no record).  A robot that simply ended its last step with no active row is neither: its empty set
    // carries its support legs, and it goes on as the reference's method does.
    build_set = s.prev_working_set != nullptr && !cold && from != stance;
  }
  // support legs first: the leg behind my slot, and what was loaded by leg goes to the lane of its slot
  const unsigned perm = slot_legs(stance);
  const int aleg = (int)((perm >> (2 * leg)) & 3u);
  // (which robot I am, where my results go and the slot order are not needed before the very end: parked like the Jacobian
  // row; a batch has fewer than 2^31 robots -- 300 bytes of state each)
  reinterpret_cast<int2 *>(lds_nrm + 64 * 11)[(int)threadIdx.x & 63] = make_int2((int)i_in, (3 * aleg + c) | ((int)perm << 8) | ((int)stance << 16));
  const bool on = leg < nS; // my slot's leg supports
  const unsigned stance_slots = (1u << nS) - 1u;
  const bool permuted = __builtin_amdgcn_ballot_w64(perm != 0xE4u) != 0ull; // (scalar: some row of the wavefront is)
  if (permuted) {
    qj = from_leg_lane(qj, aleg);
    if (kPerLeg) { nWl[0] = from_leg_lane(nWl[0], aleg); nWl[1] = from_leg_lane(nWl[1], aleg); nWl[2] = from_leg_lane(nWl[2], aleg); }
  }
#pragma unroll
  for (int j = 0; j < kTabLoads; j++) {
    const int idx = (int)threadIdx.x + kBlock * j;
    if (idx < 4 * kTabPerLeg) lds_tab[idx] = tabv[j];
  }
  __syncthreads();

  QL_STAMP(1); QL_QP_BLOCK_STAMP(4);
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
      const double s2r = 1.0 - wr_dw * wr_dw;
      // the series is evaluated inside its range only: a quaternion that is not of unit length (|d_w| > 1: the
      // reference's factor is then 2, its branch for s^2 < 1e-12) gives s^2 = 0 here, anything above the range is
      // corrected below from a bounded value
      const double s2 = vmin(vmax(s2r, 0.0), 0.01);
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
      wr_slow = !(s2r < 0.01) || !(wr_dw > 0.0);
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
  const CoopTab tab{lds_tab + kTabPerLeg * aleg};
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
  asm volatile("" ::: "memory");
  // The model table is dead from here on (one wavefront per workgroup: nobody else reads it): its place takes what only the
  // assembly of G and g0 reads -- my foot position component, the wrench of my row (replicated: every lane of the row writes the
  // same six values) -- so that it does not sit in registers across the first of the two regions below.
  static_assert(!kParkInputs || kBlock == 64, "the parked inputs take the model table's place: one wavefront per workgroup");
  if constexpr (kParkInputs) {
    lds_tab[(int)threadIdx.x & 63] = foot;
#pragma unroll
    for (int k = 0; k < 6; k++) lds_tab[64 + 6 * (((int)threadIdx.x & 63) >> 4) + k] = b[k];
  }

  QL_STAMP(3); QL_QP_BLOCK_STAMP(5);
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
  QL_STAMP(5);
  // ---------------------------------------------------------------- G row, g0 and the force QP (force_qp_coop.hpp)
  double x;
  int qp_iters;
  unsigned long long final_set = 0ull;
  // (wave-uniform: rows without a robot have no support leg)
  const bool two_legs = __builtin_amdgcn_ballot_w64(nS > 2) == 0ull;
  // Two regions one after the other, the small form first, behind flags the compiler cannot see through: as an if / else
  // the two come out in the order "12 variables, then 6" with the inputs of the second live across the first -- whose loop
  // is the one that needs every register of the 168-register form (88 bytes of scratch per lane, and 17 % on a million
  // robots).  This way the inputs of the large form are live across the small one, which has 48 registers to spare -- and
  // they are few: each region assembles its own row of G and g0 from the foot position and the wrench.
  int status = kStatusOk;
  const auto solve = [&](auto Legs) {
    ForceQp Q;
    if constexpr (kParkInputs) {
      asm volatile("" ::: "memory");
      const double foot_l = lds_tab[(int)threadIdx.x & 63];
      double b_l[6];
#pragma unroll
      for (int k = 0; k < 6; k++) b_l[k] = lds_tab[64 + 6 * (((int)threadIdx.x & 63) >> 4) + k];
      force_qp_objective(P.S, P.w_reg, foot_l, stance_slots, comp && on, b_l, nullptr, 0.0, Q.Gm, Q.g0);
    } else {
      force_qp_objective(P.S, P.w_reg, foot, stance_slots, comp && on, b, nullptr, 0.0, Q.Gm, Q.g0);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { Q.nb[k] = nb[k]; Q.t1[k] = t1[k]; Q.t2[k] = t2[k]; }
    Q.myn = myn; Q.myt1 = myt1; Q.myt2 = myt2;
    Q.mu = mu; Q.f_min = f_min;
    Q.on = on; Q.comp = comp; Q.nS = nS; Q.refine_passes = P.refine_passes;
    Q.warm = 0ull; Q.stance = stance_slots;
    Q.build_set = build_set;
    if constexpr (kWarm) Q.warm = permuted ? working_set_to_slots<5, unsigned>(warm_set, perm) : warm_set;
    status = force_qp_coop<false, kWarm, decltype(Legs)::value, !kThroughput>(Q, lds_row, lds_nrm, x, qp_iters, &final_set);
  };
  if constexpr (kSmallForm) {
    int small_form = __builtin_amdgcn_readfirstlane(two_legs ? 1 : 0);
    asm volatile("" : "+s"(small_form));
    if (small_form != 0) solve(std::integral_constant<int, 2>{});
    int large_form = __builtin_amdgcn_readfirstlane(1 - small_form);
    asm volatile("" : "+s"(large_form));
    if (large_form != 0) solve(std::integral_constant<int, 4>{});
  } else {
    (void)two_legs;
    solve(std::integral_constant<int, 4>{});
  }
  const int2 parked = reinterpret_cast<const int2 *>(lds_nrm + 64 * 11)[(int)threadIdx.x & 63];
  const int64_t i = parked.x;
  const int aidx = parked.y & 15;
  if constexpr (kWarm) {
    if (permuted) final_set = working_set_to_legs<5, unsigned>((unsigned)final_set, ((unsigned)parked.y >> 8) & 0xFFu);
  }
  if (status == kStatusNotPd) {
    if (lr == 0 && robot_live) {
      status_out[i] = kStatusNotPd;
      if (s.iterations) s.iterations[i] = 0;
      if constexpr (kWarm) { if (s.working_set) s.working_set[i] = 0u; }
    }
    if (comp && robot_live && !P.keep_on_failure && (on || !s.support_only)) { tau_out[12 * i + aidx] = 0.0; if (grf_out) grf_out[12 * i + aidx] = 0.0; }
    return false;
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
    if (comp && robot_live && !(P.keep_on_failure && status != kStatusOk) && (on || !s.support_only)) {
      tau_out[12 * i + aidx] = live ? t : 0.0;
      if (grf_out) grf_out[12 * i + aidx] = live ? x : 0.0;
    }
    if (lr == 0 && robot_live) {
      status_out[i] = status;
      if (s.iterations) s.iterations[i] = qp_iters;
      if constexpr (kWarm) { // (with the support legs it was reached with: bits 20..23)
        if (s.working_set) s.working_set[i] = status == kStatusOk ? ((uint32_t)final_set | (((uint32_t)parked.y >> 16 & 0xFu) << 20)) : 0u;
      }
    }
  }
  QL_STAMP(9);
  QL_STAMP(10);
  // (QLAMD_OPT_WARM_FALLBACK 2, diagnostics: every robot that ends with a non-empty set counts as rejected)
  const bool rejected = kWarm && robot_live && (status == kStatusWarmRejected || (P.warm_fallback == 2 && status == kStatusOk && final_set != 0ull));
  if constexpr (kWarm) {
    if (rejected && lr == 0 && s.warm_retries) atomicAdd(s.warm_retries, 1u);
  }
  return rejected;
}

// A warm start must never cost an answer (the reference ends every update() with efforts for a solvable QP,
// ros_balance_controller.cpp:418-454): a row whose warm start was rejected is solved again by its own wavefront, from the empty
// working set, before the launch ends, the other rows of the wavefront riding along empty.  The second attempt is the COLD
// instantiation of coop_robot behind a scalar branch at the very end of the kernel (its working set comes back 0: the robot's
// next step starts cold as well).  When nobody is rejected -- always, for a robot handed its own previous set -- it costs one
// ballot and a scalar branch.  With QLAMD_OPT_WARM_FALLBACK 0 the rejection is reported instead (QLAMD_STATUS_WARM_REJECTED).
// (Measured and dropped: a loop around the warm-started body -- 173 -> 256 registers and 130-600 bytes of scratch, the loops
// of the QP lose their uniform branches inside an outer loop; a call of a function of its own -- the caller's argument block
// and the callee's saved registers are 230-650 bytes of scratch per lane, which the launch pays for whether or not the call
// is ever made: 65 536 warm-started trot robots 70.7 -> 101.9 us, 4096 static +0.25 us; profiles/r6/ab_retry_forms.txt.)
// What the second attempt needs, fetched AGAIN from the kernel's argument segment through a pointer the compiler cannot see
// through: were it taken from the kernel's own parameters, every pointer of the batch would stay live across the first attempt
// (or be merged with its loads), and the 168-register form of the kernel pays for that with 200 bytes of spills in its hot
// path.  This way the only thing the cold path shares with the hot one is the argument segment's address.
template <class Args>
__device__ __forceinline__ const Args *kernel_arguments_again() {
  typedef __attribute__((address_space(4))) const char kernarg_char;
  kernarg_char *kp = (kernarg_char *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return (const Args *)kp;
}

} // namespace coop
} // namespace qlamd
