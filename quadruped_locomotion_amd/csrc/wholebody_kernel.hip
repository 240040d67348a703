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

// The dynamics entry in its throughput form: ONE LANE PER LEG, 16 robots per wavefront (wholebody_coop.hpp, wb_leg_*).
// All arithmetic first; then the outputs go through a staging block in LDS to contiguous 16-byte stores, eight robots
// (one half of the wavefront) at a time so that the block is 20 KB: M of robots 0-7, M of robots 8-15, then h and Jc
// likewise.  (270 registers: one wavefront per SIMD.  Capped at 256 it spills ten values and is no faster; what bounds
// it at 1 M robots is the store stream, 84 % of what a plain fill of the same bytes reaches.)  Every row of M and Jc a lane is responsible for is written in full (zeros
// included): no fill pass.
template <bool kM, bool kHJ>
__global__ __launch_bounds__(64) void wholebody_dynamics_leg_kernel(const DeviceParams *__restrict__ Pp, const coop::WbParamsDev W,
                                                                       const WbPtrs s, int64_t B, double *__restrict__ Mo,
                                                                       double *__restrict__ ho, double *__restrict__ Jo) {
  using namespace coop;
  __shared__ double tab[4 * kTabPerLeg];
  __shared__ __attribute__((aligned(16))) double outb[8 * kWbStage];
  const DeviceParams &P = *Pp;
  TabStage ts;
  ts.issue(P);
  const int rb = threadIdx.x >> 2, leg = threadIdx.x & 3;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int64_t i = (r0 + rb) < B ? (r0 + rb) : B - 1;
  double quat[4], linvel[3], angvel[3], q[3], qd[3];
  {
    const double2 *a2 = reinterpret_cast<const double2 *>(s.quat + 4 * i);
    double2 v = a2[0]; quat[0] = v.x; quat[1] = v.y;
    v = a2[1]; quat[2] = v.x; quat[3] = v.y;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      linvel[k] = s.linvel[3 * i + k]; angvel[k] = s.angvel[3 * i + k];
      q[k] = s.q[12 * i + 3 * leg + k]; qd[k] = s.qd[12 * i + 3 * leg + k];
    }
  }
  ts.commit(tab);
  const bool wantM = kM && Mo, wantHJ = kHJ && (ho || Jo);
  const int nrob = (int)((B - r0) < 16 ? (B - r0) : 16);
  const double2 zero2 = {0.0, 0.0};
  const int j0 = 6 + 3 * leg; // my first joint's row / column
  const int half = rb >> 3, rh = rb & 7;

  WbInertia T;                  // composite inertia of the robot
  double Fcol[3][6], Mj[3][3];  // my joints' columns of M, my leg's joint block
  WbLeg G;
  double gB[3], vB[3];
  {
    double Rm[9];
    quat_to_matrix(quat, Rm);
    const double gW[3] = {0.0, 0.0, -W.grav};
    irot(Rm, gW, gB);
    irot(Rm, linvel, vB);
    double sj[3], cj[3];
#pragma unroll
    for (int k = 0; k < 3; k++) sincos_reduced(q[k], sj[k], cj[k]);
    wb_leg_chain(CoopTab{tab + kTabPerLeg * leg}, sj, cj, G);
  }

  // ---- the mass matrix
  if (wantM) {
    wb_leg_crba(W, G, T, Fcol, Mj);
#pragma unroll 1
    for (int hf = 0; hf < 2; hf++) {
      if (half == hf) {
        double *ob = outb + kWbStage * rh;
        // the base block (the same on the four lanes of the robot), interface order [linear ; angular]:
        // [[m 1, -[h]x], [[h]x, I]], two entries per store
        {
          double2 *o2 = reinterpret_cast<double2 *>(ob);
          const double m = T.m, hx = T.h[0], hy = T.h[1], hz = T.h[2];
          const double2 r0a = {m, 0.0}, r0b = {0.0, 0.0}, r0c = {hz, -hy};
          const double2 r1a = {0.0, m}, r1b = {0.0, -hz}, r1c = {0.0, hx};
          const double2 r2a = {0.0, 0.0}, r2b = {m, hy}, r2c = {-hx, 0.0};
          const double2 r3a = {0.0, -hz}, r3b = {hy, T.I[0]}, r3c = {T.I[1], T.I[2]};
          const double2 r4a = {hz, 0.0}, r4b = {-hx, T.I[1]}, r4c = {T.I[3], T.I[4]};
          const double2 r5a = {-hy, hx}, r5b = {0.0, T.I[2]}, r5c = {T.I[4], T.I[5]};
          o2[0] = r0a; o2[1] = r0b; o2[2] = r0c;
          o2[9] = r1a; o2[10] = r1b; o2[11] = r1c;
          o2[18] = r2a; o2[19] = r2b; o2[20] = r2c;
          o2[27] = r3a; o2[28] = r3b; o2[29] = r3c;
          o2[36] = r4a; o2[37] = r4b; o2[38] = r4c;
          o2[45] = r5a; o2[46] = r5b; o2[47] = r5c;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int j = j0 + k;
          // my joint's column in the base rows, and its whole row: [force ; moment | zeros except my leg's block]
#pragma unroll
          for (int a = 0; a < 3; a++) { ob[18 * a + j] = Fcol[k][3 + a]; ob[18 * (3 + a) + j] = Fcol[k][a]; }
          double2 *rowp = reinterpret_cast<double2 *>(ob + 18 * j);
          const double2 f01 = {Fcol[k][3], Fcol[k][4]}, f2n0 = {Fcol[k][5], Fcol[k][0]}, n12 = {Fcol[k][1], Fcol[k][2]};
          rowp[0] = f01; rowp[1] = f2n0; rowp[2] = n12;
#pragma unroll
          for (int e = 3; e < 9; e++) rowp[e] = zero2;
#pragma unroll
          for (int k2 = 0; k2 < 3; k2++) ob[18 * j + j0 + k2] = k2 <= k ? Mj[k2][k] : Mj[k][k2];
        }
      }
      __syncthreads();
      const int n = nrob - 8 * hf;
      if (n > 0) copy_out(Mo + (r0 + 8 * hf) * 324, outb, 324 * (n < 8 ? n : 8));
      __syncthreads();
    }
  }
  // ---- bias forces and the contact Jacobian: h of eight robots, then their Jc
  if (wantHJ) {
    double tau[3], gb[6], Jl[3][3], pf[3];
    const double V0[6] = {angvel[0], angvel[1], angvel[2], vB[0], vB[1], vB[2]};
    const double A0[6] = {0.0, 0.0, 0.0, -gB[0], -gB[1], -gB[2]};
    const double qdd[3] = {0.0, 0.0, 0.0};
    wb_leg_inverse_dynamics(W, G, V0, A0, qd, qdd, tau, gb);
#pragma unroll
    for (int k = 0; k < 3; k++) { // J[a][k] = (z_k x (p_foot - p_k))[a]
      const double d[3] = {G.pf[0] - G.p[k][0], G.pf[1] - G.p[k][1], G.pf[2] - G.p[k][2]};
      double col[3];
      cross3(G.z[k], d, col);
#pragma unroll
      for (int a = 0; a < 3; a++) Jl[a][k] = col[a];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) pf[a] = G.pf[a];
#pragma unroll 1
    for (int hf = 0; hf < 2; hf++) {
      if (half == hf) {
        double *hb = outb + 18 * rh, *jb = outb + 8 * 18 + 216 * rh;
        double2 *h2 = reinterpret_cast<double2 *>(hb);
        const double2 g01 = {gb[0], gb[1]}, g23 = {gb[2], gb[3]}, g45 = {gb[4], gb[5]};
        h2[0] = g01; h2[1] = g23; h2[2] = g45;
#pragma unroll
        for (int k = 0; k < 3; k++) hb[j0 + k] = tau[k];
        // my leg's three rows of Jc: [1 , -[r]x | zeros except my leg's block]
        const double sk[3][3] = {{0.0, pf[2], -pf[1]}, {-pf[2], 0.0, pf[0]}, {pf[1], -pf[0], 0.0}};
#pragma unroll
        for (int a = 0; a < 3; a++) {
          double *jr = jb + 18 * (3 * leg + a);
          double2 *jr2 = reinterpret_cast<double2 *>(jr);
          const double2 e01 = {a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0}, e2s0 = {a == 2 ? 1.0 : 0.0, sk[a][0]}, s12 = {sk[a][1], sk[a][2]};
          jr2[0] = e01; jr2[1] = e2s0; jr2[2] = s12;
#pragma unroll
          for (int e = 3; e < 9; e++) jr2[e] = zero2;
#pragma unroll
          for (int k = 0; k < 3; k++) jr[j0 + k] = Jl[a][k];
        }
      }
      __syncthreads();
      const int n = nrob - 8 * hf;
      if (n > 0) {
        const int nn = n < 8 ? n : 8;
        if (ho) copy_out(ho + (r0 + 8 * hf) * 18, outb, 18 * nn);
        if (Jo) copy_out(Jo + (r0 + 8 * hf) * 216, outb + 8 * 18, 216 * nn);
      }
      __syncthreads();
    }
  }
}

// One whole-body control step per robot: inverse dynamics for the desired accelerations -> force/torque QP over the
// stance legs -> joint efforts.  The torques are eliminated through the joint rows (tau = tau0 - J_leg' f), which leaves
// 12 force variables and 11 inequality rows per stance leg (minimal normal force, friction pyramid, upper and lower
// bound of each joint torque), every one of them local to a leg: the QP runs on the quad layout the dynamics was
// computed in (lane 4 leg + c: body c of the leg = force component c = joint c), force_qp_coop.hpp with its torque rows
// -- no exchange through LDS, no general dense solver.
// kWarm: the QP starts from the working set handed in through qlamd_place_next_call (force_qp_coop.hpp; 44 rows: 64 bits).
// wholebody_robot: the step of the robot in my row (i; live: the row holds one).  Returns whether its warm start was rejected.
constexpr int kWbLdsDoubles = 4 * kTabPerLeg + 4 * coop::kCoopLdsDoubles + coop::kForceQpNrmRowsTorque * 64;
template <bool kPerLeg, bool kWarm>
__device__ __forceinline__ bool wholebody_robot(const DeviceParams &P, const coop::WbParamsDev &W, const WbPtrs &s, int64_t i, bool live,
                                                double *__restrict__ tau_out, double *__restrict__ grf_out,
                                                int32_t *__restrict__ status_out, const PlacePtrs &pp, double *lds) {
  using namespace coop;
  double *tab = lds, *rows = lds + 4 * kTabPerLeg, *nrm = rows + 4 * kCoopLdsDoubles;
  TabStage ts;
  ts.issue(P);
  const int row = threadIdx.x >> 4, lr = threadIdx.x & 15, leg = lr >> 2, c = lr & 3;
  const int jq = 3 * leg + (c < 3 ? c : 2);
  WbLaneIn in;
  in.load(s, i, jq);
  double ades[6];
#pragma unroll
  for (int k = 0; k < 6; k++) ades[k] = s.a_des[6 * i + k];
  const double qdd_raw = (s.qdd ? s.qdd : s.qd)[12 * i + jq];
  const uint32_t sm = *reinterpret_cast<const uint32_t *>(s.stance + 4 * i);
  unsigned long long warm_set = 0ull;
  bool build_set = false;
  if constexpr (kWarm) warm_set = pp.prev_working_set ? pp.prev_working_set[i] : 0ull;
  double nWl[3] = {0.0, 0.0, 1.0};
  if (kPerLeg) { nWl[0] = s.normals[12 * i + 3 * leg]; nWl[1] = s.normals[12 * i + 3 * leg + 1]; nWl[2] = s.normals[12 * i + 3 * leg + 2]; }
  ts.commit(tab);
  const unsigned stance_legs = live ? (((sm & 0xFFu) ? 1u : 0u) | ((sm & 0xFF00u) ? 2u : 0u) | ((sm & 0xFF0000u) ? 4u : 0u) |
                                       ((sm & 0xFF000000u) ? 8u : 0u))
                                    : 0u;
  const int nS = __popc(stance_legs);
  if constexpr (kWarm) { // a set remembers the support legs it was reached with (bits 44..47); other legs now: a cold start (balance_coop.hpp)
    const unsigned from = (unsigned)(warm_set >> 44) & 0xFu;
    warm_set = (from != 0u && from != stance_legs) ? 0ull : (warm_set & ((1ull << 44) - 1ull));
    build_set = pp.prev_working_set != nullptr && from != stance_legs; // (no record, or other legs: the set is built by rounds)
  }
  // support legs first (balance_coop.hpp): `leg` is my SLOT in the row, aleg the leg behind it; what was loaded by leg moves
  // to the lane of its slot
  const unsigned perm = slot_legs(stance_legs);
  const int aleg = (int)((perm >> (2 * leg)) & 3u);
  const unsigned stance = (1u << nS) - 1u;
  const bool permuted = __builtin_amdgcn_ballot_w64(perm != 0xE4u) != 0ull;
  double qdd_l = qdd_raw;
  if (permuted) {
    in.qj = from_leg_lane(in.qj, aleg); in.qdj = from_leg_lane(in.qdj, aleg); qdd_l = from_leg_lane(qdd_l, aleg);
    if (kPerLeg) { nWl[0] = from_leg_lane(nWl[0], aleg); nWl[1] = from_leg_lane(nWl[1], aleg); nWl[2] = from_leg_lane(nWl[2], aleg); }
  }
  const bool comp = c < 3, on = leg < nS, row_on = comp && on;

  // ---- dynamics on the quad lanes
  double Rm[9], gB[3], vB[3];
  quat_to_matrix(in.quat, Rm);
  const double gW[3] = {0.0, 0.0, -W.grav};
  irot(Rm, gW, gB);
  irot(Rm, in.linvel, vB);
  double sj, cj;
  sincos_reduced(in.qj, sj, cj);
  WbLink L;
  wb_link(CoopTab{tab + kTabPerLeg * aleg}, c, sj, cj, L);
  const double V0[6] = {in.angvel[0], in.angvel[1], in.angvel[2], vB[0], vB[1], vB[2]};
  const double A0[6] = {ades[3], ades[4], ades[5], ades[0] - gB[0], ades[1] - gB[1], ades[2] - gB[2]};
  double tau0, gb[6];
  wb_inverse_dynamics(W, L, c, V0, A0, comp ? in.qdj : 0.0, (comp && s.qdd) ? qdd_l : 0.0, tau0, gb);
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
  Q.warm = 0ull; Q.stance = stance; Q.build_set = build_set;
  if constexpr (kWarm) Q.warm = permuted ? working_set_to_slots<11, unsigned long long>(warm_set, perm) : warm_set;
  int qp_iters;
  unsigned long long final_set = 0ull;
  // the two forms of the QP as two regions behind opaque flags, the 6-variable one first (balance_coop.hpp)
  const bool two_legs = __builtin_amdgcn_ballot_w64(nS > 2) == 0ull;
  int st = kStatusOk;
  int small_form = __builtin_amdgcn_readfirstlane(two_legs ? 1 : 0);
  asm volatile("" : "+s"(small_form));
  if (small_form != 0) st = force_qp_coop<true, kWarm, 2>(Q, rows + kCoopLdsDoubles * row, nrm, x, qp_iters, &final_set);
  int large_form = __builtin_amdgcn_readfirstlane(1 - small_form);
  asm volatile("" : "+s"(large_form));
  if (large_form != 0) st = force_qp_coop<true, kWarm, 4>(Q, rows + kCoopLdsDoubles * row, nrm, x, qp_iters, &final_set);
  if constexpr (kWarm) {
    if (permuted) final_set = working_set_to_legs<11, unsigned long long>(final_set, perm);
  }

  // ---- joint efforts: tau = tau0 - J_leg' f on the stance legs, tau0 elsewhere
  const bool ok = st == kStatusOk;
  const double f = (row_on && ok) ? x : 0.0;
  const double tq = tau0 - (Q.jcol[0] * quad_bc<0>(f) + Q.jcol[1] * quad_bc<1>(f) + Q.jcol[2] * quad_bc<2>(f));
  // a failed robot: zeros, or (QLAMD_ON_FAILURE_KEEP) its entries stay as the caller left them
  if (comp && live && !(P.keep_on_failure && !ok)) {
    tau_out[12 * i + 3 * aleg + c] = ok ? tq : 0.0;
    if (grf_out) grf_out[12 * i + 3 * aleg + c] = f;
  }
  if (lr == 0 && live) {
    status_out[i] = st;
    if (pp.iterations) pp.iterations[i] = st == kStatusNotPd ? 0 : qp_iters;
    if constexpr (kWarm) { if (pp.working_set) pp.working_set[i] = st == kStatusOk ? (final_set | ((unsigned long long)stance_legs << 44)) : 0ull; }
  }
  bool rejected = false;
  if constexpr (kWarm) {
    rejected = live && (st == kStatusWarmRejected || (P.warm_fallback == 2 && st == kStatusOk && final_set != 0ull));
    if (rejected && lr == 0 && pp.warm_retries) atomicAdd(pp.warm_retries, 1u);
  }
  return rejected;
}

// The second attempt of the rows whose warm start was rejected: the cold step as a function of its own that ends the wavefront and
// fetches the kernel's arguments again (balance_kernel.hip, balance_cold_retry, has the reasons).
struct WbSolveArgs { const DeviceParams *Pp; coop::WbParamsDev W; WbPtrs s; int64_t B; double *tau, *grf; int32_t *status; PlacePtrs pp; };
template <bool kPerLeg>
__device__ __attribute__((noinline, noreturn)) void wholebody_cold_retry(const WbSolveArgs *args, double *lds, bool rejected) {
  const WbSolveArgs &a = *args;
  bool inside;
  const int64_t i = placed_index(a.pp, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 4), a.B, inside);
  const PlacePtrs cold{a.pp.order, a.pp.iterations, nullptr, nullptr, nullptr};
  (void)wholebody_robot<kPerLeg, false>(*a.Pp, a.W, a.s, i, rejected, a.tau, a.grf, a.status, cold, lds);
  if (rejected && (threadIdx.x & 15) == 0 && a.pp.working_set) a.pp.working_set[i] = 0ull;
  __builtin_amdgcn_endpgm();
}

template <bool kPerLeg, bool kWarm = false>
__global__ __launch_bounds__(64, 2) void wholebody_solve_kernel(const DeviceParams *__restrict__ Pp, const coop::WbParamsDev W,
                                                             const WbPtrs s, int64_t B, double *__restrict__ tau_out,
                                                             double *__restrict__ grf_out, int32_t *__restrict__ status_out,
                                                             const PlacePtrs pp) {
  __shared__ double lds[kWbLdsDoubles];
  bool live;
  const int64_t i = placed_index(pp, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 4), B, live);
  const bool rejected = wholebody_robot<kPerLeg, kWarm>(*Pp, W, s, i, live, tau_out, grf_out, status_out, pp, lds);
  if constexpr (kWarm) {
    // a warm start must never cost an answer: the rejected rows are solved again, cold, by this wavefront (balance_coop.hpp)
    if (__builtin_expect(Pp->warm_fallback && __builtin_amdgcn_ballot_w64(rejected) != 0ull, 0)) {
      __syncthreads();
      wholebody_cold_retry<kPerLeg>(coop::kernel_arguments_again<WbSolveArgs>(), lds, rejected);
    }
  }
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
  // Two layouts of the same arithmetic.  Measured (profiles/r2): the row form 9.8 / 13.9 / 22.5 us at 4096 / 8192 / 16 384
  // robots against 13.1 / 14.3 / 19.8 for the leg form, which then pulls away (1 M robots: 0.92 ms against 1.22 ms).
  const bool leg_form = ctx->dynamics_form == QLAMD_DYNAMICS_LEG || (ctx->dynamics_form == QLAMD_DYNAMICS_AUTO && batch > 8192);
  if (leg_form) {
    // one lane per leg, 16 robots per wavefront; M, then h and Jc, from one launch
    const dim3 grid((unsigned)((batch + 15) / 16));
    if (dM && (dh || dJ))
      hipLaunchKernelGGL((wholebody_dynamics_leg_kernel<true, true>), grid, dim3(64), 0, st, ctx->d_params, W, s, batch, dM, dh, dJ);
    else if (dM)
      hipLaunchKernelGGL((wholebody_dynamics_leg_kernel<true, false>), grid, dim3(64), 0, st, ctx->d_params, W, s, batch, dM,
                         (double *)nullptr, (double *)nullptr);
    else
      hipLaunchKernelGGL((wholebody_dynamics_leg_kernel<false, true>), grid, dim3(64), 0, st, ctx->d_params, W, s, batch,
                         (double *)nullptr, dh, dJ);
  } else {
    // 16 lanes per robot, 4 robots per wavefront
    const dim3 grid((unsigned)((batch + 3) / 4));
    if (dM && (dh || dJ)) {
      hipLaunchKernelGGL((wholebody_dynamics_kernel<true, true>), grid, dim3(64), 0, st, ctx->d_params, W, s, batch, dM, dh, dJ);
    } else {
      if (dM)
        hipLaunchKernelGGL((wholebody_dynamics_kernel<true, false>), grid, dim3(64), 0, st, ctx->d_params, W, s, batch, dM,
                           (double *)nullptr, (double *)nullptr);
      if (dh || dJ)
        hipLaunchKernelGGL((wholebody_dynamics_kernel<false, true>), grid, dim3(64), 0, st, ctx->d_params, W, s, batch,
                           (double *)nullptr, dh, dJ);
    }
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
  PlacePtrs pp;
  qlamd_placement pl;
  { const int rc = take_placement(ctx, memory, batch, &pp, &pl); if (rc != QLAMD_OK) return rc; }
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
    const bool keep = ctx->params.keep_on_failure != 0; // entries the kernel leaves alone come back as they went up
    sg.add(joint_effort, B * 96, keep, true);
    sg.add(contact_force, B * 96, keep, true);
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
  const bool warm = pp.prev_working_set || pp.working_set;
#define QL_LAUNCH_WB(PERLEG)                                                                                                   \
  do {                                                                                                                         \
    if (warm)                                                                                                                  \
      hipLaunchKernelGGL((wholebody_solve_kernel<PERLEG, true>), dim3(grid), dim3(64), 0, st, ctx->d_params, W, s, batch, dtau, \
                         dgrf, dst, pp);                                                                                       \
    else                                                                                                                       \
      hipLaunchKernelGGL((wholebody_solve_kernel<PERLEG, false>), dim3(grid), dim3(64), 0, st, ctx->d_params, W, s, batch,     \
                         dtau, dgrf, dst, pp);                                                                                 \
  } while (0)
  if (s.normals) QL_LAUNCH_WB(true);
  else QL_LAUNCH_WB(false);
#undef QL_LAUNCH_WB
  if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  { const int rc = finish_placement(ctx, pl, batch, st); if (rc != QLAMD_OK) return rc; }
  if (memory == QLAMD_MEM_HOST) return sg.finish(st);
  return QLAMD_OK;
}

} // extern "C"

QLAMD_STAMPS_ACCESSOR(qlamd_debug_stamps_wholebody)
