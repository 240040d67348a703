// HIP kernels (gfx950) of the steps either side of the balance solve in a control tick (rows a18, f1, f2, leg IK of
// f4): swing-leg torque and swing branch, leg state machine, message unpacking, analytic leg IK, and the whole tick
// composed of them; with their part of the C-ABI of include/qlamd.h.
#define QL_QP_BLOCK_STAMP(slot) // slots 1 and 2 of this unit: the end of an unpack block, the start of a solve block
#include "balance_coop.hpp"
#include "swing_core.hpp"
#include "leg_state_core.hpp"
#include "wire_core.hpp"
#include "context.hpp"

using namespace qlamd;
using namespace qlamd::rt;

namespace {

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
  const uint8_t *live; // [B] or NULL (whole tick): 0 = robot left alone
};

// One (robot, leg) per lane; `block`: index of the 64-lane block among the swing blocks; tab: 4 * kTabPerLeg doubles of LDS.
__device__ __forceinline__ void swing_branch_block(const DeviceParams &P, const SwingParamsDev &SP, const PidParamsDev &pid,
                                                   const SwingPtrs &s, const SwingBranchPtrs &b, double period, int64_t B,
                                                   double *__restrict__ effort, int64_t block, double *tab) {
  TabStage ts;
  ts.issue(P);
  const int64_t t0 = block * 64 + threadIdx.x;
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
  const bool alive = !b.live || b.live[i] != 0;
  PidLeg pl; // this leg's gains out of the kernel arguments, fetched with everything else
  pid_leg_of(pid, leg, pl);
  ts.commit(tab);
  if (support || !live || !alive) return; // support legs keep the clamped QP torque already in `effort` (:497-502)
  double out[3];
  swing_branch_leg(LdsTab{tab + kTabPerLeg * leg}, SP, pl, b.mode ? mode : 0, quat, qi, q, qd, qo, tp, tv, cmd, period, el,
                   ei, out);
#pragma unroll
  for (int k = 0; k < 3; k++) { effort[3 * t + k] = out[k]; b.e_last[3 * t + k] = el[k]; b.e_int[3 * t + k] = ei[k]; }
}

__global__ __launch_bounds__(64) void swing_branch_kernel(const DeviceParams *__restrict__ Pp, const SwingParamsDev SP,
                                                          const PidParamsDev pid, const SwingPtrs s,
                                                          const SwingBranchPtrs b, double period, int64_t B,
                                                          double *__restrict__ effort) {
  __shared__ double tab[4 * kTabPerLeg];
  swing_branch_block(*Pp, SP, pid, s, b, period, B, effort, (int64_t)blockIdx.x, tab);
}

// The whole tick's two solvers in ONE launch: blocks [0, nbal) are the balance step (balance_coop.hpp: 4 robots per
// block, the efforts of the support legs), the blocks behind them the swing branch (64 (robot, leg) pairs per block,
// the efforts of the other legs).  Neither reads what the other writes.  The balance blocks are dispatched first and
// occupy one wavefront per SIMD for as long as their slowest robot needs; the swing blocks run in that shadow, so
// their 8 us and a launch gap disappear from the tick.  (Two streams joined by events cost as much as they saved.)
struct TickSwingArgs {
  SwingParamsDev SP;
  PidParamsDev pid;
  SwingPtrs s;
  SwingBranchPtrs b;
  double period;
};
// The second attempt of the balance rows whose warm start was rejected: the cold step as a function of its own that ends the
// wavefront and fetches the kernel's arguments again (balance_kernel.hip, balance_cold_retry, has the reasons).
struct TickSolveArgs { const DeviceParams *Pp; coop::CoopPtrs cp; int64_t B; double *effort; int32_t *status; }; // the kernel's first parameters
__device__ __attribute__((noinline, noreturn)) void tick_cold_retry(const TickSolveArgs *args, double *tab, double *rows, double *nrm, bool rejected) {
  const TickSolveArgs &a = *args;
  const int row = threadIdx.x >> 4;
  int64_t ir = (int64_t)blockIdx.x * 4 + row;
  if (ir >= a.B) ir = a.B - 1;
  coop::CoopPtrs cold = a.cp;
  cold.prev_working_set = nullptr; cold.working_set = nullptr; cold.warm_retries = nullptr;
  (void)coop::coop_robot<false, 64, false>(*a.Pp, cold, ir, rejected, tab, rows + row * coop::kCoopLdsDoubles, nrm, a.effort, nullptr, a.status);
  if (rejected && (threadIdx.x & 15) == 0 && a.cp.working_set) a.cp.working_set[ir] = 0u;
  __builtin_amdgcn_endpgm();
}
template <bool kWarm>
__global__ __launch_bounds__(64, 2) void tick_solve_kernel(const DeviceParams *__restrict__ Pp, const coop::CoopPtrs cp, int64_t B,
                                                           double *__restrict__ effort, int32_t *__restrict__ status,
                                                           unsigned nbal, const TickSwingArgs sw) {
  __shared__ double tab[4 * kTabPerLeg];
  __shared__ double rows[4 * coop::kCoopLdsDoubles];
  __shared__ double nrm[coop::kCoopNrmDoubles];
  QL_BLOCK_STAMP(2);
  if (blockIdx.x < nbal) {
    const int row = threadIdx.x >> 4;
    int64_t i = (int64_t)blockIdx.x * 4 + row;
    const bool live = i < B;
    if (!live) i = B - 1;
    const bool rejected = coop::coop_robot<false, 64, kWarm>(*Pp, cp, i, live, tab, rows + row * coop::kCoopLdsDoubles, nrm, effort, nullptr, status);
    if constexpr (kWarm) {
      // a rejected warm start is solved again cold by the same wavefront (balance_kernel.hip, balance_coop_kernel)
      if (__builtin_expect(Pp->warm_fallback && __builtin_amdgcn_ballot_w64(rejected) != 0ull, 0)) {
        __syncthreads();
        tick_cold_retry(coop::kernel_arguments_again<TickSolveArgs>(), tab, rows, nrm, rejected);
      }
    }
  } else {
    swing_branch_block(*Pp, sw.SP, sw.pid, sw.s, sw.b, sw.period, B, effort, (int64_t)(blockIdx.x - nbal), tab);
  }
  QL_BLOCK_STAMP(3);
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
  // the whole tick only: this tick's mode names as decoded from the message and the modes in force (in/out); a known
  // name replaces the mode in force, anything else leaves it (:876-964), and is_footstep is derived from the result
  const uint8_t *msg_mode;
  uint8_t *leg_mode;
  // the whole tick only: robots without a command in force (live == 0) are left alone and reported in status
  const uint8_t *live;
  int32_t *status;
};

// Everything the state machine of robot i reads, fetched in one go (independent loads, one round trip): the four flags of a
// kind travel as one 32-bit word.  Split from the machine itself so that the parser's blocks can issue these loads at their
// very start -- they do not depend on the message -- and take the command fields of a well-formed message straight from the
// record in LDS instead of reading them back from the arrays they have just written (two dependent memory round trips at the
// tail of every block of the unpack launch: 12.2 -> 8.6 us without them, profiles/r5/tick_block_stamps.txt).
struct LegStateIn {
  uint32_t sup, fst_in, mm, mcur, con, lst, sto, sup_i;
  double ph[4], jp[12], sj[12], ft[12];
  uint8_t live;
};
__device__ __forceinline__ void leg_state_load(const LegStatePtrs &s, int64_t i, LegStateIn &in) {
  in.sup = *reinterpret_cast<const uint32_t *>(s.support_leg + 4 * i);
  // (the whole tick hands in this tick's mode names instead of is_footstep: both words are loaded here with everything
  // else, merged below, and the modes in force are written back with the other results)
  in.fst_in = *reinterpret_cast<const uint32_t *>((s.msg_mode ? s.support_leg : s.is_footstep) + 4 * i);
  in.mm = *reinterpret_cast<const uint32_t *>((s.msg_mode ? s.msg_mode : s.support_leg) + 4 * i);
  in.mcur = *reinterpret_cast<const uint32_t *>((s.msg_mode ? (const uint8_t *)s.leg_mode : s.support_leg) + 4 * i);
  in.con = *reinterpret_cast<const uint32_t *>(s.contact + 4 * i);
  in.lst = *reinterpret_cast<const uint32_t *>(s.limb_state + 4 * i);
  in.sto = *reinterpret_cast<const uint32_t *>(s.store_flag + 4 * i);
  const double2 p01 = *reinterpret_cast<const double2 *>(s.phase + 4 * i);
  const double2 p23 = *reinterpret_cast<const double2 *>(s.phase + 4 * i + 2);
  in.ph[0] = p01.x; in.ph[1] = p01.y; in.ph[2] = p23.x; in.ph[3] = p23.y;
  // every array the tick may touch is fetched up front (independent 16-byte loads, one round trip); what
  // the state machine decides only selects which values are written back
  {
    const double2 *pj = reinterpret_cast<const double2 *>(s.joint_position + 12 * i);
    const double2 *ps = reinterpret_cast<const double2 *>(s.stored_joint_position + 12 * i);
    const double2 *pf = reinterpret_cast<const double2 *>(s.foot_target + 12 * i);
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double2 a = pj[k], b = ps[k], c = pf[k];
      in.jp[2 * k] = a.x; in.jp[2 * k + 1] = a.y; in.sj[2 * k] = b.x; in.sj[2 * k + 1] = b.y; in.ft[2 * k] = c.x; in.ft[2 * k + 1] = c.y;
    }
  }
  in.sup_i = *reinterpret_cast<const uint32_t *>(s.support + 4 * i);
  in.live = s.live ? s.live[i] : (uint8_t)1;
}

// the state machine of robot i on what leg_state_load fetched, by one lane
__device__ __forceinline__ void leg_state_run(const LegStatePtrs &s, int index_quirk, int64_t i, const LegStateIn &in) {
  LegStateRobot r;
  const uint32_t sup = in.sup, mm = in.mm, mcur = in.mcur, con = in.con, lst = in.lst, sto = in.sto;
  const double *ph = in.ph, *jp = in.jp, *sj = in.sj, *ft = in.ft;
  uint32_t fst = in.fst_in, merged = 0;
  if (s.msg_mode) {
    fst = 0;
#pragma unroll
    for (int l = 0; l < 4; l++) {
      const uint32_t m = (mm >> (8 * l)) & 0xFFu, c = (mcur >> (8 * l)) & 0xFFu;
      const uint32_t v = m != (uint32_t)kModeOther ? m : c;
      merged |= v << (8 * l);
      fst |= (v == (uint32_t)kModeFootstep ? 1u : 0u) << (8 * l);
    }
  }
#pragma unroll
  for (int l = 0; l < 4; l++) {
    r.support_leg[l] = ((sup >> (8 * l)) & 0xFFu) != 0;
    r.is_footstep[l] = ((fst >> (8 * l)) & 0xFFu) != 0;
    r.contact[l] = ((con >> (8 * l)) & 0xFFu) != 0;
    r.limb_state[l] = (int)(int8_t)((lst >> (8 * l)) & 0xFFu);
    r.store_flag[l] = ((sto >> (8 * l)) & 0xFFu) != 0;
    r.phase[l] = ph[l];
  }
  const uint32_t sup_i = in.sup_i;
  if (s.live && !in.live) {
    s.status[i] = QLAMD_STATUS_NO_COMMAND;
    return;
  }
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
  if (s.msg_mode) *reinterpret_cast<uint32_t *>(s.leg_mode + 4 * i) = merged;
}

__device__ __forceinline__ void leg_state_robot(const LegStatePtrs &s, int index_quirk, int64_t i) {
  LegStateIn in;
  leg_state_load(s, i, in);
  leg_state_run(s, index_quirk, i, in);
}

__global__ __launch_bounds__(256) void leg_state_kernel(const LegStatePtrs s, int index_quirk, int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < B) leg_state_robot(s, index_quirk, i);
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
// The chain pos -> length field -> pos is what the walk costs (one LDS round trip and a handful of dependent
// instructions per field, one lane), so a step is kept to the minimum: `field(pre)` steps over `pre` fixed bytes, reads
// the uint32 there and leaves pos behind it; `over(n, post)` steps over n variable and `post` fixed bytes.  Only the
// read position and the result of `over` are clamped to cap = len + 1 (the staging window reaches 16 bytes past the last
// message, so a read at cap is harmless): once a field points past the end pos sticks at cap ("bad"), the status is
// decided by pos >= cap at the very end, and anchors noted on the way are only used for a message that ended well.
// kLog: this message's (position, value) pairs become the next launch's layout template.
template <bool kLog>
struct WireSkeleton {
  const LdsBytes &p;
  uint32_t pos, cap;
  uint32_t *log;
  uint32_t nf;
  __device__ __forceinline__ uint32_t field(uint32_t pre) {
    const uint32_t at = min(pos + pre, cap);
    const uint32_t v = p.u32(at);
    if (kLog) {
      if (nf < (uint32_t)kTplMaxFields) { log[2 * nf] = at; log[2 * nf + 1] = v; }
    }
    nf++;
    pos = at + 4u;
    return v;
  }
  __device__ __forceinline__ void over(uint32_t n, uint32_t post) { pos = min(pos + min(n, cap) + post, cap); }
  __device__ __forceinline__ bool bad() const { return pos >= cap; }
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
template <bool kLog>
__device__ __forceinline__ int wire_lds_skeleton(const LdsBytes &src, int64_t len64, uint32_t *an, uint32_t *log, uint32_t &nf,
                                                 uint32_t &end_pos) {
  nf = 0u; end_pos = 0u;
  if (len64 < 0 || len64 > 0x7FFFFFF0ll) return kWireTruncated;
  const uint32_t len = (uint32_t)len64;
  WireSkeleton<kLog> c{src, 0u, len + 1u, log, 0u};
  bool missing = false;
  const auto times8 = [](uint32_t n) { return n > 0x0FFFFFFFu ? 0xFFFFFFFFu : 8u * n; };
#pragma nounroll
  for (int l = 0; l < 4; l++) { // sensor_msgs/JointState
    c.over(c.field(12), 0);                       // header: seq, stamp, frame_id
    uint32_t nn = c.field(0);
#pragma nounroll
    for (; nn > 0 && !c.bad(); nn--) c.over(c.field(0), 0);
    const uint32_t np = c.field(0);
    missing = missing || np < 3;
    an[kAnJointPos + l] = c.pos;
    an[kAnJointCnt + l] = np;
    c.over(times8(np), 0);
    c.over(times8(c.field(0)), 0);
    c.over(times8(c.field(0)), 0);
  }
  c.over(c.field(12), 0);                         // nav_msgs/Odometry: header
  c.over(c.field(0), 0);                          // child_frame_id
  an[kAnOdomPose] = c.pos;
  c.over(0, 56 + 288 + 48 + 288);
#pragma nounroll
  for (int l = 0; l < 4; l++) {                   // free_gait_msgs/LegMode
    const uint32_t n = c.field(0);
    an[kAnModeName + l] = c.pos;
    an[kAnModeLen + l] = n;
    c.over(n, 0);
    an[kAnModeFlag + l] = c.pos;
    c.over(c.field(1 + 8 + 8 + 12), 0);           // support_leg, duration, phase; header
    an[kAnModeNormal + l] = c.pos;
    c.over(0, 24 + 1);
  }
#pragma nounroll
  for (int l = 0; l < 4; l++) {                   // free_gait_msgs/EndEffectorTarget
    c.over(c.field(0), 0);                        // name
#pragma nounroll
    for (int arr = 0; arr < 4; arr++) {
      const uint32_t n = c.field(0);
      if (arr < 3) missing = missing || n == 0;
#pragma nounroll
      for (uint32_t k = 0; k < n && !c.bad(); k++) {
        c.over(c.field(12), 24);                  // header; the 24 payload bytes
        if (k == 0 && arr < 3) an[kAnTarget + 3 * l + arr] = c.pos - 24u;
      }
    }
    c.over(c.field(8 + 12), 24 + 2);              // average_velocity; header; surface_normal.vector, two flags
  }
  nf = c.nf; end_pos = c.pos;
  if (c.bad()) return kWireTruncated; // some field ran past the end: the record stays cleared
  return missing ? kWireMissingField : kWireOk;
}

// Pass 2: payload at the anchors, by the 16 lanes of the message's row, straight to the output arrays (robot i).  One step per
// output array, lane lr = the array's entry lr of the robot (the four short ones of the base -- des_pos 0-2, des_quat 3-6,
// des_linvel 7-9, des_angvel 10-12 -- share a step): which array a step reads for and stores to is known at compile time, a
// lane's address in it is a constant plus 8 lr -- no per-entry dispatch on either side, and no trip through a record in LDS
// (round 6; before: five entries lr + 16 t of a 77-entry record per lane, written to LDS and written out from there by ten
// loops behind a barrier: extraction 1.4 + write-out 1.6 us of a block's 10).  In two halves, so that the reads can be issued
// at the TEMPLATE's anchors together with the reads of the template check, before the check has decided (a row whose message
// then turns out to have another layout reads again, at the anchors its walk found).
struct WireRowAnchors { // the anchors lane lr of a row needs: leg l3 = lr / 3 of the 12-wide arrays, leg l4 = lr & 3 of the per-leg ones
  uint32_t odom, joint, n_joint, t0, t1, t2, nrm, flag, name, n_name;
};
struct WireRowPayload {
  double base, joint, t0, t1, t2, nrm, phase;
  uint32_t w0, w1, w2;
  uint8_t sup_byte;
};
// get(slot): anchor `slot` of the message (0 for a message without them)
template <class Get>
__device__ __forceinline__ WireRowAnchors wire_row_anchors(const Get &get, int lr) {
  const int l3 = lr < 12 ? lr / 3 : 3, l4 = lr & 3;
  return WireRowAnchors{get(kAnOdomPose), get(kAnJointPos + l3), get(kAnJointCnt + l3), get(kAnTarget + 3 * l3), get(kAnTarget + 3 * l3 + 1),
                        get(kAnTarget + 3 * l3 + 2), get(kAnModeNormal + l3), get(kAnModeFlag + l4), get(kAnModeName + l4), get(kAnModeLen + l4)};
}
// Byte source in global memory: reads at any byte offset (the hardware takes unaligned global loads), every address clamped so
// that the read stays inside the message.
struct GlobalBytes {
  const uint8_t *p;
  uint32_t len; // >= 16
  __device__ __forceinline__ uint32_t u32(uint32_t at) const { uint32_t v; __builtin_memcpy(&v, p + min(at, len - 4u), 4); return v; }
  __device__ __forceinline__ uint8_t u8(uint32_t at) const { return p[min(at, len - 1u)]; }
  __device__ __forceinline__ double f64(uint32_t at) const { double v; __builtin_memcpy(&v, p + min(at, len - 8u), 8); return v; }
};
template <class Bytes>
__device__ __forceinline__ WireRowPayload wire_row_read(const Bytes &src, const WireRowAnchors &a, int lr) {
  const uint32_t j8 = 8u * (uint32_t)(lr - 3 * (lr / 3));
  // base step: wire order of the pose is position, orientation (x, y, z, w) -> (w, x, y, z); the twist follows 288 bytes of covariance
  const uint32_t off0 = lr < 3 ? 8u * lr : lr < 7 ? 24u + 8u * (lr & 3) : lr < 13 ? 56u + 288u + 8u * (lr - 7) : 0u;
  WireRowPayload p;
  p.base = src.f64(a.odom + off0);
  p.joint = src.f64(a.joint + j8);
  p.t0 = src.f64(a.t0 + j8); p.t1 = src.f64(a.t1 + j8); p.t2 = src.f64(a.t2 + j8);
  p.nrm = src.f64(a.nrm + j8);
  p.phase = src.f64(a.flag + 9u);
  p.w0 = src.u32(a.name); p.w1 = src.u32(a.name + 4); p.w2 = src.u32(a.name + 8);
  p.sup_byte = src.u8(a.flag);
  return p;
}
// have: the message was walked (or matched the template) to its end -- otherwise every entry is the cleared record's 0.
// write: a well-formed message, or the parse-only entry, which reports every record.  What the state machine at the block's
// tail reads of the command (support flags, modes, phases, foot positions) is also left in the record f in LDS.
__device__ __forceinline__ void wire_row_store(const WireRowPayload &p, const WireRowAnchors &a, RobotStateFields &f, int lr, bool have,
                                               const RobotStateOutPtrs &out, int64_t i, bool write) {
  const uint32_t j3 = (uint32_t)(lr - 3 * (lr / 3));
  // "joint" 5, "leg_mode" 8, "cartesian" 9, "footstep" 8: compare 12 bytes read as three words against the literals
  int mode = kModeOther;
  if (a.n_name == 5 && p.w0 == 0x6E696F6Au && (p.w1 & 0xFFu) == 0x74u) mode = kModeJoint;                      // "join" "t"
  else if (a.n_name == 8 && p.w0 == 0x5F67656Cu && p.w1 == 0x65646F6Du) mode = kModeLegMode;                  // "leg_" "mode"
  else if (a.n_name == 9 && p.w0 == 0x74726163u && p.w1 == 0x61697365u && (p.w2 & 0xFFu) == 0x6Eu) mode = kModeCartesian; // "cart" "esia" "n"
  else if (a.n_name == 8 && p.w0 == 0x746F6F66u && p.w1 == 0x70657473u) mode = kModeFootstep;                 // "foot" "step"
  mode = have ? mode : (int)kModeOther;
  const uint8_t sup = have && p.sup_byte != 0;
  const double e_base = have ? p.base : 0.0, e_joint = have && j3 < a.n_joint ? p.joint : 0.0, e_t0 = have && a.t0 != 0u ? p.t0 : 0.0,
               e_t1 = have && a.t1 != 0u ? p.t1 : 0.0, e_t2 = have && a.t2 != 0u ? p.t2 : 0.0, e_nrm = have ? p.nrm : 0.0,
               e_phase = have ? p.phase : 0.0;
  const bool w12 = lr < 12;
  if (w12) f.foot_position[lr] = e_t0;
  if (lr < 4) { f.phase[lr] = e_phase; f.leg_mode[lr] = (uint8_t)mode; f.support_leg[lr] = sup; }
  if (!write) return;
  {
    double *b = lr < 3 ? out.des_pos : lr < 7 ? out.des_quat : lr < 10 ? out.des_linvel : out.des_angvel;
    const int64_t k = lr < 3 ? 3 * i + lr : lr < 7 ? 4 * i + (lr - 3) : 3 * i + (lr < 10 ? lr - 7 : lr - 10);
    if (lr < 13 && b) b[k] = e_base;
  }
  const int64_t k12 = 12 * i + lr;
  if (w12 && out.joint_command) out.joint_command[k12] = e_joint;
  if (w12 && out.foot_position) out.foot_position[k12] = e_t0;
  if (w12 && out.foot_velocity) out.foot_velocity[k12] = e_t1;
  if (w12 && out.foot_acceleration) out.foot_acceleration[k12] = e_t2;
  if (w12 && out.surface_normal) out.surface_normal[k12] = e_nrm;
  if (lr < 4) {
    if (out.phase) out.phase[4 * i + lr] = e_phase;
    if (out.support_leg) out.support_leg[4 * i + lr] = sup;
    if (out.leg_mode) out.leg_mode[4 * i + lr] = (uint8_t)mode;
  }
}

constexpr int kWireMsgsPerBlock = 4;          // messages parsed per 64-lane block (one 16-lane row each)
constexpr int kWireLdsBytes = 32 * 1024;      // staging window; longer runs are parsed straight from global memory
// The window of a launch that is bound by throughput, not by a block's latency: 19 KB of LDS a block instead of 35, eight blocks
// on a CU instead of four -- a tick of 65 536 robots 211 -> 186 us, of 16 384: 80 -> 71 (profiles/r6/README.md).  Four messages of up to 4 KB fit (a
// reference message is 3.3 KB); for device buffers, whose sizes the host does not see, it is used from kWireSmallWindowBatch
// messages on -- below that a CU holds four blocks at most anyway.  For host buffers the window is the smallest that holds
// every block's run.
constexpr int kWireLdsBytesSmall = 16 * 1024;
constexpr int64_t kWireSmallWindowBatch = 8192;
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
// The kernel's parameters as they lie in its argument segment: the extraction and the state machine at a block's tail fetch
// their pointers from there again (coop::kernel_arguments_again) -- 29 pointers kept in scalar registers from the block's first
// instruction were spilled to vector lanes and read back one v_readlane at a time (141 of the tail's 850 instructions).
struct UnpackArgs {
  const uint8_t *messages; const int64_t *offsets; int64_t B; RobotStateOutPtrs o; int32_t *status; const uint32_t *tpl_in;
  uint32_t *tpl_out; uint8_t *valid; LegStatePtrs ls; int leg_state_mode; int window_bytes;
};
__global__ __launch_bounds__(64) void robot_state_unpack_kernel(const uint8_t *__restrict__ messages,
                                                                const int64_t *__restrict__ offsets, int64_t B,
                                                                const RobotStateOutPtrs o, int32_t *__restrict__ status,
                                                                const uint32_t *__restrict__ tpl_in,
                                                                uint32_t *__restrict__ tpl_out,
                                                                uint8_t *__restrict__ valid, const LegStatePtrs ls,
                                                                int leg_state_mode, int window_bytes) {
  // leg_state_mode (whole tick): 0 = parse only; 1 / 2 = the block also runs the leg state machine of its robots on the
  // command now in force (without / with the reference's index quirk): one launch and one memory round trip less per tick
  // valid != NULL (whole tick): the outputs are the per-robot command in force -- only a well-formed message
  // replaces a robot's record and sets valid[robot]; a malformed one leaves both as they are.
  extern __shared__ uint32_t wire_lds[];
  __shared__ int okm[kWireMsgsPerBlock], via_rec[kWireMsgsPerBlock];
  __shared__ RobotStateFields rec[kWireMsgsPerBlock];
  __shared__ uint32_t anchors[kWireMsgsPerBlock][kAnCount];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * kWireMsgsPerBlock;
  const int n = (int)((B - i0) < kWireMsgsPerBlock ? (B - i0) : kWireMsgsPerBlock);
  QL_STAMP(20);
  QL_BLOCK_STAMP(0);
  const int64_t a = offsets[i0], b = offsets[i0 + n];
  const int row = tid >> 4, lr = tid & 15;
  const bool mine = row < n;
  // (my row's own pair of offsets with the block's: one round trip for all four)
  const int64_t ma = offsets[i0 + (mine ? row : 0)], mb = offsets[i0 + (mine ? row : 0) + 1];
  // the whole tick: what the state machine of my robot reads (lane m < n: robot i0 + m) is fetched NOW, with the block's first
  // loads -- none of it depends on the message; the command fields among it are the command still in force, which a well-formed
  // message replaces below, from the record in LDS
  LegStateIn lin;
  if (leg_state_mode && tid < n) leg_state_load(ls, i0 + tid, lin);
  // The template goes from memory straight into the registers of the lanes that use it, with the block's first loads: lane lr of
  // every row checks length fields lr, lr + 16, ... (a (position, value) pair each) and carries anchors lr, lr + 16, lr + 32;
  // the four header words are the same for everybody.  (Through LDS -- stored behind the staging loop, read back by the check --
  // the check waited one more LDS round trip: 1.24 -> 0.84 us, profiles/r6/tick_block_phases.txt.)
  const uint32_t tpl_valid = tpl_in[kTplValid], tpl_end = tpl_in[kTplEnd], tpl_missing = tpl_in[kTplMissing], tnf = tpl_in[kTplFields];
  uint2 pair[kTplMaxFields / 16];
#pragma unroll
  for (int t = 0; t < kTplMaxFields / 16; t++) { pair[t].x = tpl_in[kTplPairs + 2 * (lr + 16 * t)]; pair[t].y = tpl_in[kTplPairs + 2 * (lr + 16 * t) + 1]; }
  // ... and the template's anchors of the slots this lane extracts for (wire_row_anchors)
  const WireRowAnchors tpl_anchor = wire_row_anchors([&](int slot) { return tpl_in[kTplAnchors + slot]; }, lr);
  const bool sane = mine && !(ma < a || mb > b || mb < ma); // offsets not ascending: nothing to parse
  // ---- template check, 16 lanes per message, STRAIGHT FROM MEMORY (round 6): a message whose length fields hold the template's
  // values at the template's positions needs nothing else of its bytes than the payload at the template's anchors -- so the lanes
  // read exactly those, length fields and payload in one round trip, and a block whose four messages all match never stages
  // anything (before: every block copied its messages into LDS first -- a second dependent round trip plus the LDS one of the
  // check; a block 8.4 -> 7.6 us, the launch 10.0-10.7 -> 9.5: profiles/r6/tick_block_phases.txt).  Every address is clamped into the message: whatever a
  // template holds, nothing outside the message is read.
  bool same = sane && tpl_valid == kTplMagic && tnf <= (uint32_t)kTplMaxFields && (mb - ma) <= 0x7FFFFFF0ll &&
              (uint64_t)(mb - ma) >= (uint64_t)tpl_end && tpl_end >= 16u;
  WireRowAnchors anc = tpl_anchor;
  {
    const auto in = [&](uint32_t &v, bool position) { v = same ? (position ? min(v, tpl_end) : v) : 0u; };
    in(anc.odom, true); in(anc.joint, true); in(anc.n_joint, false); in(anc.t0, true); in(anc.t1, true); in(anc.t2, true);
    in(anc.nrm, true); in(anc.flag, true); in(anc.name, true); in(anc.n_name, false);
  }
  WireRowPayload pay{};
  if (same) {
    const GlobalBytes gsrc{messages + ma, (uint32_t)(mb - ma)};
    // eight length fields per lane and the lane's share of the payload, all reads independent
    uint32_t got[kTplMaxFields / 16];
#pragma unroll
    for (int t = 0; t < kTplMaxFields / 16; t++) got[t] = gsrc.u32(lr + 16u * t < tnf ? pair[t].x : 0u);
    pay = wire_row_read(gsrc, anc, lr);
#pragma unroll
    for (int t = 0; t < kTplMaxFields / 16; t++) same = same && (lr + 16u * t >= tnf || got[t] == pair[t].y);
  }
  const bool hit = ((unsigned)(__ballot(same) >> (tid & 48)) & 0xFFFFu) == 0xFFFFu;
  const bool all_hit = __ballot(mine && !hit) == 0ull; // (wavefront-uniform) nobody has to be walked: nothing is staged
  QL_STAMP(21); QL_BLOCK_STAMP(4);
  // ---- a block with a message of another layout (or no template yet): its messages are staged in LDS for the walk
  const uintptr_t src = (uintptr_t)(messages + a);
  const uintptr_t src_al = src & ~(uintptr_t)15;
  const int64_t lead = (int64_t)(src - src_al), nbytes = lead + (b - a);
  const bool staged = !all_hit && nbytes + 16 <= window_bytes; // +16: u32 / f64 reads may touch the next two words
  if (!all_hit) {
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
  // clear the records (fields a malformed message never reaches read as zero)
  if (!staged)
    for (int w = tid; w < (int)(sizeof(rec) / 4); w += 64) ((uint32_t *)rec)[w] = 0u;
  for (int w = tid; w < kWireMsgsPerBlock * kAnCount; w += 64) (&anchors[0][0])[w] = 0u;
  __syncthreads();
  }
  const LdsBytes msg{wire_lds, (uint32_t)(lead + (ma - a))};
  QL_STAMP(22); QL_BLOCK_STAMP(5);
  uint32_t nf = 0u, end_pos = 0u;
  int st = kWireTruncated;
  const bool logger = blockIdx.x == 0 && row == 0; // this message's layout becomes the next launch's template
  if (lr == 0 && mine) {
    if (!sane) st = kWireTruncated;
    else if (hit) st = tpl_missing ? kWireMissingField : kWireOk;
    else if (staged && logger) st = wire_lds_skeleton<true>(msg, mb - ma, anchors[row], tpl_out + kTplPairs, nf, end_pos);
    else if (staged) st = wire_lds_skeleton<false>(msg, mb - ma, anchors[row], nullptr, nf, end_pos);
    else {
      st = robot_state_unpack(PlainBytes{messages + ma}, mb - ma, rec[row]);
      if (st == kWireTruncated) // (the fields met before the end: a truncated message delivers the cleared record, as when staged)
        for (int w = 0; w < (int)(sizeof(RobotStateFields) / 4); w++) ((uint32_t *)&rec[row])[w] = 0u;
    }
    status[i0 + row] = st;
    okm[row] = (st == kWireOk || !valid) ? 1 : 0;
    via_rec[row] = (!hit && !staged && sane) ? okm[row] : 0; // parsed into the record by one lane: written out from there below
    if (valid && st == kWireOk) valid[i0 + row] = 1;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const int st_row = __shfl(st, 0, 16);
  // a row stores its entries straight from the registers of the extraction (a malformed message of the parse-only entry: the
  // cleared record); the record in LDS is for the state machine below and for a message parsed from global memory by one lane
  if (mine && (hit || staged || !sane)) {
    const bool have = sane && st_row != kWireTruncated;
    if (!hit) { // another layout (or no template yet): the anchors the walk left in LDS
      const uint32_t *an = anchors[row];
      anc = wire_row_anchors([&](int slot) { return have ? an[slot] : 0u; }, lr);
      pay = wire_row_read(LdsBytes{wire_lds, have ? msg.shift : 0u}, anc, lr);
    }
    wire_row_store(pay, anc, rec[row], lr, have, coop::kernel_arguments_again<UnpackArgs>()->o, i0 + row, st_row == kWireOk || !valid);
  }
  QL_STAMP(23);
  // ---- the template for the next launch (block 0, first message).  Its body goes out here; its valid word goes last,
  // behind a fence (a template is never valid before all of it is in memory) -- at the very END of the block's work, where
  // the fence finds the body's stores long complete: placed here it held block 0, and with it the launch, up for 2 us.
  uint32_t tpl_valid_word = 0u;
  if (logger) {
    const uint32_t nf0 = __shfl(nf, 0, 16), end0 = __shfl(end_pos, 0, 16);
    if (hit) {
      tpl_valid_word = tpl_valid; // still in force: the body is copied below, by the whole block
    } else {
      // the walk has already left its (position, value) pairs in tpl_out; valid only if the message was well-formed
      const bool good = sane && staged && st_row != kWireTruncated && nf0 <= (uint32_t)kTplMaxFields;
      if (lr == 0) {
        tpl_out[kTplEnd] = end0;
        tpl_out[kTplMissing] = st_row == kWireMissingField ? 1u : 0u; tpl_out[kTplFields] = nf0;
      }
      for (int k = lr; k < kAnCount; k += 16) tpl_out[kTplAnchors + k] = anchors[0][k];
      tpl_valid_word = good ? kTplMagic : 0u;
    }
  }
  if (blockIdx.x == 0 && __builtin_amdgcn_readfirstlane((int)hit) != 0) { // row 0 hit: all 64 lanes copy the template
    for (int k = tid; k < kTplWords; k += 64)
      if (k != kTplValid) tpl_out[k] = tpl_in[k];
  }
  __syncthreads();
  QL_STAMP(24); QL_BLOCK_STAMP(6);
  // write-out: message m's k doubles of each field are contiguous in the output arrays
  const auto put = [&](double *dst, int width, size_t field_off) {
    if (!dst || staged || all_hit) return;
    for (int e = tid; e < n * width; e += 64) {
      const int m = e / width, k2 = e - m * width;
      if (via_rec[m]) dst[(int64_t)width * i0 + e] = ((const double *)((const char *)&rec[m] + field_off))[k2];
    }
  };
  put(o.des_pos, 3, offsetof(RobotStateFields, des_pos)); put(o.des_quat, 4, offsetof(RobotStateFields, des_quat));
  put(o.des_linvel, 3, offsetof(RobotStateFields, des_linvel)); put(o.des_angvel, 3, offsetof(RobotStateFields, des_angvel));
  put(o.joint_command, 12, offsetof(RobotStateFields, joint_command));
  put(o.foot_position, 12, offsetof(RobotStateFields, foot_position));
  put(o.foot_velocity, 12, offsetof(RobotStateFields, foot_velocity));
  put(o.foot_acceleration, 12, offsetof(RobotStateFields, foot_acceleration));
  put(o.surface_normal, 12, offsetof(RobotStateFields, surface_normal)); put(o.phase, 4, offsetof(RobotStateFields, phase));
  if (tid < 4 * n && !staged && !all_hit) {
    const int m = tid >> 2, l = tid & 3;
    if (o.support_leg && via_rec[m]) o.support_leg[4 * i0 + tid] = rec[m].support_leg[l];
    if (o.leg_mode && via_rec[m]) o.leg_mode[4 * i0 + tid] = rec[m].leg_mode[l];
  }
  QL_STAMP(25); QL_BLOCK_STAMP(7);
  if (leg_state_mode) {
    __syncthreads(); // (the block's records are on their way to memory before the state machine writes over parts of them)
    if (tid < n) {
      if (okm[tid]) { // this tick's message is the command in force: its fields straight from the record
        const RobotStateFields &r = rec[tid];
        lin.sup = lin.fst_in = *reinterpret_cast<const uint32_t *>(r.support_leg);
        lin.mm = *reinterpret_cast<const uint32_t *>(r.leg_mode);
#pragma unroll
        for (int k = 0; k < 4; k++) lin.ph[k] = r.phase[k];
#pragma unroll
        for (int k = 0; k < 12; k++) lin.ft[k] = r.foot_position[k];
        lin.live = 1;
      }
      leg_state_run(coop::kernel_arguments_again<UnpackArgs>()->ls, leg_state_mode == 2, i0 + tid, lin);
    }
  }
  if (logger) {
    __threadfence();
    if (lr == 0) tpl_out[kTplValid] = tpl_valid_word;
  }
  QL_BLOCK_STAMP(1);
}

} // namespace

extern "C" {

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
  QL_ENTER(ctx, st);
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

} // extern "C"

// live: device pointer [B] or NULL (whole tick, QLAMD_MEM_DEVICE only): robots with 0 are left alone
static int swing_branch_impl(qlamd_context *ctx, const qlamd_swing_params *params, const qlamd_joint_pid_params *pid,
                             const qlamd_swing_batch *in, const qlamd_swing_branch_extra *extra, const uint8_t *live,
                             double period, int64_t batch, double *joint_effort, int memory, void *stream) {
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
  QL_ENTER(ctx, st);
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
                     extra->pid_error_integral, live};
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
                         sg.dev<double>(b4), nullptr};
    d_eff = sg.dev<double>(e0);
  }
  hipLaunchKernelGGL(swing_branch_kernel, dim3((unsigned)((4 * batch + 63) / 64)), dim3(64), 0, st, ctx->d_params, SP, PD,
                     s, sb, period, batch, d_eff);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  return memory == QLAMD_MEM_HOST ? sg.finish(st) : QLAMD_OK;
}

extern "C" {

int qlamd_swing_branch_batch(qlamd_context *ctx, const qlamd_swing_params *params, const qlamd_joint_pid_params *pid,
                             const qlamd_swing_batch *in, const qlamd_swing_branch_extra *extra, double period,
                             int64_t batch, double *joint_effort, int memory, void *stream) {
  return swing_branch_impl(ctx, params, pid, in, extra, nullptr, period, batch, joint_effort, memory, stream);
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
  QL_ENTER(ctx, st);
  const size_t B = (size_t)batch;
  LegStatePtrs s{io->support_leg, io->is_footstep, io->contact, io->phase, io->joint_position, io->limb_state,
                 io->store_flag, io->stored_joint_position, io->joint_command, io->foot_target, io->support,
                 io->leg_state_code, nullptr, nullptr, nullptr, nullptr};
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
                     sg.dev<double>(9), sg.dev<uint8_t>(10), sg.dev<int8_t>(11), nullptr, nullptr, nullptr, nullptr};
  }
  hipLaunchKernelGGL(leg_state_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, s, index_quirk, batch);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

} // extern "C"

// valid: device pointer [B] or NULL (whole tick, QLAMD_MEM_DEVICE only), see robot_state_unpack_kernel
static int unpack_impl(qlamd_context *ctx, const uint8_t *messages, const int64_t *offsets, int64_t batch,
                       const qlamd_robot_state_fields *out, int32_t *status, uint8_t *valid, int memory, void *stream,
                       const LegStatePtrs *ls = nullptr, int leg_state_mode = 0) {
  if (!ctx || !messages || !offsets || !out || !status || batch < 0) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  QL_ENTER(ctx, st);
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
    if (rt::CallGuard::capturing(st)) return QLAMD_ERR_NEEDS_RESERVE;
    if (hipMalloc((void **)&ctx->wire_tpl, 2 * kTplWords * sizeof(uint32_t)) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
    if (hipMemsetAsync(ctx->wire_tpl, 0, 2 * kTplWords * sizeof(uint32_t), st) != hipSuccess) return QLAMD_ERR_HIP;
  }
  // Launches read one half of the template block and leave the next launch's template in the other, and the host flips
  // -- for eager launches.  A captured launch is replayed with the pointers it was captured with, so it reads the half in
  // force at capture time on every replay and nothing is flipped (the capture may be discarded): the template the graph
  // reads is the one the last EAGER launch left.  Run one eager tick before capturing (include/qlamd.h); without one the
  // graph's template is "none" and every replay walks every message -- slower, never wrong.  (Reading and writing one half
  // instead would let a block that starts late see block 0's new field list with the old extraction anchors.)
  const bool captured = rt::CallGuard::capturing(st);
  const uint32_t *tpl_in = ctx->wire_tpl + kTplWords * ctx->wire_flip;
  uint32_t *tpl_out = ctx->wire_tpl + kTplWords * (ctx->wire_flip ^ 1);
  if (!captured) ctx->wire_flip ^= 1;
  int window = batch >= kWireSmallWindowBatch ? kWireLdsBytesSmall : kWireLdsBytes;
  if (memory == QLAMD_MEM_HOST) { // the host sees the offsets: the smaller window if every block's run fits it
    window = kWireLdsBytesSmall;
    for (int64_t i = 0; i < batch && window == kWireLdsBytesSmall; i += kWireMsgsPerBlock) {
      const int64_t j = i + kWireMsgsPerBlock < batch ? i + kWireMsgsPerBlock : batch;
      if (offsets[j] - offsets[i] + 32 > kWireLdsBytesSmall) window = kWireLdsBytes; // (+16 of alignment lead, +16 of overread)
    }
  }
  hipLaunchKernelGGL(robot_state_unpack_kernel, dim3((unsigned)((batch + kWireMsgsPerBlock - 1) / kWireMsgsPerBlock)),
                     dim3(64), window, st, d_msg, d_off, batch, o, d_st, tpl_in, tpl_out, valid,
                     ls ? *ls : LegStatePtrs{}, ls ? leg_state_mode : 0, window);
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

extern "C" {

int qlamd_robot_state_unpack_batch(qlamd_context *ctx, const uint8_t *messages, const int64_t *offsets, int64_t batch,
                                   const qlamd_robot_state_fields *out, int32_t *status, int memory, void *stream) {
  return unpack_impl(ctx, messages, offsets, batch, out, status, nullptr, memory, stream);
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
  QL_ENTER(ctx, st);
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

// Layout of the command block of the whole tick (qlamd_tick_batch.command): per-robot flag "a command is in force",
// then what the last well-formed message delivered, one array per field like the intermediates they are.
namespace {
enum { kCmdValid, kCmdPos, kCmdQuat, kCmdLin, kCmdAng, kCmdJoint, kCmdFootP, kCmdFootV, kCmdPhase, kCmdSup, kCmdMode, kCmdN };
inline size_t command_layout(size_t B, size_t off[kCmdN]) {
  const size_t sz[kCmdN] = {B, B * 24, B * 32, B * 24, B * 24, B * 96, B * 96, B * 96, B * 32, B * 4, B * 4};
  size_t total = 0;
  for (int k = 0; k < kCmdN; k++) { off[k] = total; total += align256(sz[k]); }
  return total;
}
} // namespace

size_t qlamd_tick_command_bytes(int64_t batch) {
  size_t off[kCmdN];
  return batch > 0 ? command_layout((size_t)batch, off) : 0;
}

int qlamd_reserve(qlamd_context *ctx, int64_t max_batch) {
  if (!ctx || max_batch < 1) return QLAMD_ERR_INVALID_ARGUMENT;
  QL_ENTER_NO_STREAM(ctx);
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  size_t coff[kCmdN];
  const size_t scratch = align256((size_t)max_batch * 4) + command_layout((size_t)max_batch, coff);
  if (ctx->tick_ws_bytes < scratch) {
    // Growing means a device synchronisation and a free, neither of which is legal while a stream is being captured (the
    // header says so: reserve BEFORE the capture).  The runtime refuses the synchronisation then; that refusal is
    // reported as what it is rather than as a HIP failure, and nothing has been freed.
    {
      const hipError_t e = hipDeviceSynchronize(); // earlier calls may still use the old block
      if (e == hipErrorStreamCaptureUnsupported || e == hipErrorStreamCaptureImplicit || e == hipErrorStreamCaptureInvalidated) {
        (void)hipGetLastError();
        return QLAMD_ERR_NEEDS_RESERVE;
      }
      if (e != hipSuccess) return QLAMD_ERR_HIP;
    }
    if (ctx->tick_ws) (void)hipFree(ctx->tick_ws);
    ctx->tick_ws = nullptr; ctx->tick_ws_bytes = 0;
    if (hipMalloc(&ctx->tick_ws, scratch) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
    ctx->tick_ws_bytes = scratch;
  }
  if (!ctx->wire_tpl) {
    if (hipMalloc((void **)&ctx->wire_tpl, 2 * kTplWords * sizeof(uint32_t)) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
    if (hipMemset(ctx->wire_tpl, 0, 2 * kTplWords * sizeof(uint32_t)) != hipSuccess) return QLAMD_ERR_HIP;
  }
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
  QL_ENTER(ctx, st);
  const size_t B = (size_t)batch;
  size_t coff[kCmdN];
  const size_t cmd_bytes = command_layout(B, coff);
  qlamd_tick_batch d = *io;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    for (size_t k = 0; k < B; k++)
      if (io->offsets[k + 1] < io->offsets[k] || io->offsets[0] < 0) return QLAMD_ERR_INVALID_ARGUMENT;
    const size_t nbytes = (size_t)(io->offsets[B] - io->offsets[0]);
    const int i_off = sg.add(io->offsets, (B + 1) * 8, true, false);
    const int i_msg = sg.add(io->messages + io->offsets[0], nbytes, true, false); // an empty blob stages nothing
    const int i_in[8] = {sg.add(io->joint_position, B * 96, true, false), sg.add(io->joint_velocity, B * 96, true, false),
                         sg.add(io->joint_velocity_oldest, B * 96, true, false), sg.add(io->base_position, B * 24, true, false),
                         sg.add(io->base_orientation, B * 32, true, false), sg.add(io->base_linear_velocity, B * 24, true, false),
                         sg.add(io->base_angular_velocity, B * 24, true, false), sg.add(io->contact, B * 4, true, false)};
    const int i_io[7] = {sg.add(io->limb_state, B * 4, true, true), sg.add(io->store_flag, B * 4, true, true),
                         sg.add(io->stored_joint_position, B * 96, true, true), sg.add(io->leg_mode, B * 4, true, true),
                         sg.add(io->pid_error_last, B * 96, true, true), sg.add(io->pid_error_integral, B * 96, true, true),
                         sg.add(io->support, B * 4, true, true)};
    // the efforts travel both ways: robots that are skipped, or whose solve fails under QLAMD_ON_FAILURE_KEEP, keep theirs
    const int i_out[4] = {sg.add(io->joint_effort, B * 96, true, true), sg.add(io->leg_state_code, B * 4, false, true),
                          sg.add(io->status, B * 4, false, true), sg.add(io->message_status, B * 4, false, true)};
    const int i_cmd = sg.add(io->command, cmd_bytes, true, true);
    const int i_ws = sg.add(io->working_set, B * 4, true, true);
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
    d.command = sg.dev<char>(i_cmd);
    d.working_set = sg.dev<uint32_t>(i_ws);
  }
  // context scratch: the leg state codes when the caller does not want them, and the command block when the caller
  // keeps none (then no command outlives the call: the flags are cleared first)
  const size_t scratch = align256(B * 4) + (d.command ? 0 : cmd_bytes);
  if (ctx->tick_ws_bytes < scratch) {
    if (rt::CallGuard::capturing(st)) return QLAMD_ERR_NEEDS_RESERVE;
    if (ctx->tick_ws) (void)hipFree(ctx->tick_ws);
    ctx->tick_ws = nullptr; ctx->tick_ws_bytes = 0;
    if (hipMalloc(&ctx->tick_ws, scratch) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
    ctx->tick_ws_bytes = scratch;
  }
  char *w = (char *)(d.command ? d.command : (char *)ctx->tick_ws + align256(B * 4));
  if (!d.command && hipMemsetAsync(w + coff[kCmdValid], 0, B, st) != hipSuccess) return QLAMD_ERR_HIP;
  const auto D = [&](int k) { return (double *)(w + coff[k]); };
  const auto U = [&](int k) { return (uint8_t *)(w + coff[k]); };
  int rc;
  // 1. baseCommandCallback: a well-formed message replaces the robot's command in force (desired state, targets, modes)
  qlamd_robot_state_fields f{};
  f.des_pos = D(kCmdPos); f.des_quat = D(kCmdQuat); f.des_linvel = D(kCmdLin); f.des_angvel = D(kCmdAng);
  f.joint_command = D(kCmdJoint); f.foot_position = D(kCmdFootP); f.foot_velocity = D(kCmdFootV); f.phase = D(kCmdPhase);
  f.support_leg = U(kCmdSup); f.leg_mode = U(kCmdMode);
  const uint8_t *live = U(kCmdValid);
  // 2. (same launch) leg modes in force, footContactsCallback + the switch of update(): support legs, held joint
  //    commands, nudged foot targets -- the state machine of qlamd_leg_state_machine_batch with the mode merge in front,
  //    run by the parser's blocks on the command they have just put in force; robots without a command get their status
  //    here and are left alone by every kernel of the tick
  const LegStatePtrs ls{U(kCmdSup), nullptr, d.contact, D(kCmdPhase), d.joint_position, d.limb_state, d.store_flag,
                        d.stored_joint_position, D(kCmdJoint), D(kCmdFootP), d.support,
                        d.leg_state_code ? d.leg_state_code : (int8_t *)ctx->tick_ws, U(kCmdMode), d.leg_mode, live, d.status};
  // (large batches: its own launch, one robot per lane, 256 per block -- the parser is bandwidth-bound there and four
  // busy lanes at the tail of every block cost more than a launch: 402 against 365 us at 65 536 robots)
  const bool one_launch = batch <= 16384;
  rc = unpack_impl(ctx, d.messages, d.offsets, batch, &f, d.message_status, U(kCmdValid), QLAMD_MEM_DEVICE, stream,
                   one_launch ? &ls : nullptr, index_quirk ? 2 : 1);
  if (rc != QLAMD_OK) return rc;
  if (!one_launch) {
    hipLaunchKernelGGL(leg_state_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, ls, index_quirk, batch);
    if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  }
  // 3. + 4. the balance solve for the support legs and the swing branch for the others
  qlamd_swing_params sp = *swing;
  sp.period = period;
  if (pick_rpw(ctx, batch) == 4 && batch <= 16384) {
    // one launch (tick_solve_kernel); beyond a few wavefronts per SIMD the swing blocks no longer find idle issue slots
    // and pay for the balance kernel's registers and LDS instead (65 536 robots: 381 us fused, 365 us apart)
    if (!(sp.period > 0.0) || !(sp.accel_window > 0.0)) return QLAMD_ERR_INVALID_ARGUMENT;
    TickSwingArgs ta;
    for (int k = 0; k < 3; k++) { ta.SP.kp[k] = sp.kp[k]; ta.SP.kd[k] = sp.kd[k]; }
    ta.SP.period = sp.period; ta.SP.accel_window = sp.accel_window; ta.SP.accel_scale = sp.accel_scale;
    ta.SP.gravity = sp.gravity;
    memcpy(ta.pid.p, pid->p, sizeof(ta.pid.p)); memcpy(ta.pid.i, pid->i, sizeof(ta.pid.i)); memcpy(ta.pid.d, pid->d, sizeof(ta.pid.d));
    memcpy(ta.pid.i_max, pid->i_max, sizeof(ta.pid.i_max)); memcpy(ta.pid.i_min, pid->i_min, sizeof(ta.pid.i_min));
    memcpy(ta.pid.lower, pid->lower, sizeof(ta.pid.lower)); memcpy(ta.pid.upper, pid->upper, sizeof(ta.pid.upper));
    ta.pid.antiwindup = pid->antiwindup;
    ta.s = SwingPtrs{d.joint_position, d.joint_velocity, d.joint_velocity_oldest, D(kCmdFootP), D(kCmdFootV), nullptr, d.support};
    ta.b = SwingBranchPtrs{d.base_orientation, D(kCmdJoint), d.leg_mode, d.pid_error_last, d.pid_error_integral, live};
    ta.period = period;
    // (working_set: the robot's final working set of its previous tick in, this tick's out -- in place: a robot's set is read
    // and written by its own 16 lanes only)
    const coop::CoopPtrs cp{d.joint_position, d.base_position, d.base_orientation, d.base_linear_velocity,
                            d.base_angular_velocity, D(kCmdPos), D(kCmdQuat), D(kCmdLin), D(kCmdAng), d.support,
                            nullptr, nullptr, live, 1, nullptr, d.working_set, d.working_set,
                            (uint32_t *)ctx->place_sync + kSyncWarmRetries};
    const unsigned nbal = (unsigned)((batch + 3) / 4), nsw = (unsigned)((4 * batch + 63) / 64);
    if (d.working_set)
      hipLaunchKernelGGL(tick_solve_kernel<true>, dim3(nbal + nsw), dim3(64), 0, st, ctx->d_params, cp, batch, d.joint_effort,
                         d.status, nbal, ta);
    else
      hipLaunchKernelGGL(tick_solve_kernel<false>, dim3(nbal + nsw), dim3(64), 0, st, ctx->d_params, cp, batch, d.joint_effort,
                         d.status, nbal, ta);
    if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  } else {
    // two launches (large batches; the one-lane balance kernels of qlamd_set_robots_per_wave, cross-checks)
    qlamd_state_batch sb{d.joint_position, d.base_position, d.base_orientation, d.base_linear_velocity, d.base_angular_velocity,
                         D(kCmdPos), D(kCmdQuat), D(kCmdLin), D(kCmdAng), d.support, nullptr};
    qlamd_placement wp;
    memset(&wp, 0, sizeof(wp));
    const bool coop = pick_rpw(ctx, batch) == 4; // (the one-lane kernels of the cross-check know neither warm start nor placement)
    const bool warm = d.working_set && coop;
    if (warm) wp.prev_working_set = wp.working_set = d.working_set; // in place
    const bool placed = io->placement_state && memory == QLAMD_MEM_DEVICE && coop;
    if (placed) {
      // the caller's loop of include/qlamd.h on the tick's own state: tick k runs in order[k & 1] (identity on the first tick),
      // writes iters[k & 1] and makes order[(k + 1) & 1] from iters[(k - 1) & 1] (zeros on the first tick: the identity)
      int32_t *ps = io->placement_state;
      if (ctx->tick_place_state != ps || ctx->tick_place_batch != batch) {
        ctx->tick_place_state = ps; ctx->tick_place_batch = batch; ctx->tick_place_count = 0;
        if (hipMemsetAsync(ps + 2 * B, 0, 2 * B * sizeof(int32_t), st) != hipSuccess) return QLAMD_ERR_HIP;
      }
      const int64_t k = ctx->tick_place_count++;
      wp.robot_order = k == 0 ? nullptr : ps + (k & 1) * B;
      wp.iterations = ps + 2 * B + (k & 1) * B;
      wp.prev_iterations = ps + 2 * B + ((k + 1) & 1) * B;
      wp.next_robot_order = ps + ((k + 1) & 1) * B;
      wp.policy = QLAMD_PLACEMENT_AUTO;
    }
    rc = balance_impl(ctx, &sb, nullptr, live, 1, batch, d.joint_effort, nullptr, d.status, QLAMD_MEM_DEVICE, stream,
                      (warm || placed) ? &wp : nullptr);
    if (rc != QLAMD_OK) return rc;
    const qlamd_swing_batch sw{d.joint_position, d.joint_velocity, d.joint_velocity_oldest, D(kCmdFootP), D(kCmdFootV), d.support, nullptr};
    const qlamd_swing_branch_extra ex{d.base_orientation, D(kCmdJoint), d.leg_mode, d.pid_error_last, d.pid_error_integral};
    rc = swing_branch_impl(ctx, &sp, pid, &sw, &ex, live, period, batch, d.joint_effort, QLAMD_MEM_DEVICE, stream);
    if (rc != QLAMD_OK) return rc;
  }
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

} // extern "C"

QLAMD_STAMPS_ACCESSOR(qlamd_debug_stamps_tick)
QLAMD_BLOCK_STAMPS_ACCESSOR(qlamd_debug_block_stamps_tick)
