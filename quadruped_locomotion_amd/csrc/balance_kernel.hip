// HIP kernels (gfx950) of the batched balance-controller step and their part of the
// C-ABI of include/qlamd.h, plus the context.  See DESIGN.md for the data layout and launch geometry.
#include "balance_coop.hpp"
#include "params_build.hpp"
#include "context.hpp"

using namespace qlamd;
using namespace qlamd::rt;

namespace {

struct StatePtrs {
  const double *q, *pos, *quat, *linvel, *angvel, *dpos, *dquat, *dlinvel, *dangvel;
  const uint8_t *stance;
  const double *normals;
  const double *wrench; // [B][6] or NULL
  const uint8_t *live;  // [B] or NULL: 0 = robot left alone (nothing written)
  int support_only;     // whole tick: only the support legs' efforts are written (the swing branch, which writes the
                        // others, runs beside this kernel on another stream)
  const int32_t *order; // [B] or NULL: slot s of the launch (row s % 4 of wavefront s / 4) takes robot order[s]
  int32_t *iterations;  // [B] or NULL: outer iterations of each robot's QP
  // the placement of the NEXT launch, made by extra wavefronts in the shadow of this one (placement_wave below)
  const int32_t *prev_iterations; // [B]: the counts it is made from (the previous launch's `iterations`)
  int32_t *next_order;            // [B] or NULL: where it goes.  With shadow_blocks == 0: QLAMD_PLACEMENT_NONE -- every slot writes its own
                                  // index here (no field of its own: one more pointer in the arguments cost the 168-register form 1 % at 65 536 robots)
  int place_throughput;           // policy: 0 latency, 1 throughput
  const uint32_t *prev_working_set; // [B] or NULL: warm start (kWarm instantiations)
  uint32_t *working_set;            // [B] or NULL
  int shadow_blocks, shadow_chunk;  // workgroups in front of the launch that make next_order, and the robots each of them takes
  uint32_t *place_hist;             // [shadow_blocks][kPlaceKeys]: their counts per key, for each other (the context's scratch)
  uint32_t *place_sync;             // two words, zero when the context is created: arrivals, state (their barrier)
  uint32_t place_wait;              // polls they wait for each other before they give up (QLAMD_OPT_PLACEMENT_WAIT)
  uint32_t *warm_retries;           // the context's count of rejected warm starts (kWarm instantiations)
  int record_doubles;               // 0, or the record length of QLAMD_STATE_RECORDS (lane-cooperative kernels, device memory)
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
    if (i < B && (!s.live || s.live[i])) {
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
    if (lane < RPW && i < B && (!s.live || s.live[i])) {
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
    if (i < B && (!s.live || s.live[i])) {
      LdsScratch scr{scratch + rb, RPW};
      if (P.keep_on_failure && scr.at(kScrStatus) != 0.0) continue;
      if (s.support_only && s.stance[4 * i + leg] == 0) continue;
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

// Placement of the robots into the slots of the next launch from the iteration counts of the last one
// (qlamd_placement_from_iterations): a stable counting sort by iteration count, hardest first (ties by robot index), then
// the slot of each rank by policy:
//   throughput  slot = rank: four neighbours of the sorted list share a wavefront.  The union of four similar add / drop
//               sequences is the shortest there is, which is what counts once every SIMD holds several wavefronts, and the
//               long wavefronts start first.
//   latency     the hardest quarter one per wavefront (row 0 of wavefront r = rank r), each joined by the three easiest
//               robots still to be had (rank B-1-e sits in row 1 + e % 3 of wavefront e / 3): a launch of one wavefront
//               per SIMD lasts as long as its slowest wavefront, and next to three robots that finish early a hard robot
//               runs at the speed it has alone (finished rows ride along as ghost rows, force_qp_coop.hpp).
// It sits between two control steps of a caller that wants the hint used at once, so it is built for latency: one
// workgroup of 1024 lanes takes 1024 G consecutive robots, robot g * 1024 + lane in its round g.  A robot's ordinal among
// the robots of its wavefront and round with the same count is the return value of ONE LDS atomic on the counter
// [bin][round][wavefront] (the lanes of one instruction that meet on an address are served in lane order); an exclusive
// scan over the 512 G counters in (bin, round, wavefront) order turns them into first ranks; rank -> slot -> one store.
// Batches beyond 4096 robots take several workgroups, which need the counts of the other workgroups per bin: a first
// launch (placement_hist_kernel) leaves them in the context's scratch.
constexpr int kPlaceThreads = 1024, kPlaceWaves = kPlaceThreads / 64, kPlaceBins = 24, kPlaceRounds = 4;
__device__ __forceinline__ int place_bin(int v) { // bin 0 = hardest (the clamp compiles to one v_med3_i32)
  const int h = v < 0 ? 0 : (v >= kPlaceBins ? kPlaceBins - 1 : v);
  return kPlaceBins - 1 - h;
}
// The launches of their own also know a robot's CLASS when the caller's support flags are at hand (`support`: [B][4] bytes
// read as one word per robot; NULL: one class): robots on more than two legs first, then the robots on at most two --
// whose wavefronts then take the 6-variable form of the QP (balance_coop.hpp, "support legs first": a wavefront takes it
// only when all four of its robots can).  Key = class * kPlaceBins + bin; sorted placement only (placing the hardest
// robots one per wavefront mixes the classes by design).
constexpr int kPlaceKeys = 2 * kPlaceBins;
__device__ __forceinline__ int place_key(int v, const uint32_t *__restrict__ support, int64_t i) {
  int key = place_bin(v);
  if (support) {
    const uint32_t w = support[i];
    const int n = ((w & 0xFFu) ? 1 : 0) + ((w & 0xFF00u) ? 1 : 0) + ((w & 0xFF0000u) ? 1 : 0) + ((w & 0xFF000000u) ? 1 : 0);
    key += n <= 2 ? kPlaceBins : 0;
  }
  return key;
}
__global__ __launch_bounds__(kPlaceThreads) void placement_hist_kernel(const int32_t *__restrict__ iters, int64_t B, int64_t per_block,
                                                                      const uint32_t *__restrict__ support,
                                                                      uint32_t *__restrict__ blockhist) {
  __shared__ uint32_t h[kPlaceKeys];
  if (threadIdx.x < kPlaceKeys) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * per_block, hi = lo + per_block < B ? lo + per_block : B;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kPlaceThreads) atomicAdd(&h[place_key(iters[i], support, i)], 1u);
  __syncthreads();
  if (threadIdx.x < kPlaceKeys) blockhist[(int64_t)blockIdx.x * kPlaceKeys + threadIdx.x] = h[threadIdx.x];
}
template <int G>
__global__ __launch_bounds__(kPlaceThreads) void placement_kernel(const int32_t *__restrict__ iters, int64_t B, int throughput,
                                                                 const uint32_t *__restrict__ support,
                                                                 const uint32_t *__restrict__ blockhist, int32_t *__restrict__ order) {
  constexpr int kN = kPlaceKeys * G * kPlaceWaves, kPer = (kN + kPlaceThreads - 1) / kPlaceThreads; // counters, counters per lane in the scan
  __shared__ uint32_t cnt[kN];
  __shared__ uint32_t wtot[kPlaceWaves];
  __shared__ uint32_t bin_base[kPlaceKeys];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t first = (int64_t)blockIdx.x * (kPlaceThreads * G);
  int v[G];
#pragma unroll
  for (int g = 0; g < G; g++) { // all loads in flight before the first atomic
    const int64_t i = first + g * kPlaceThreads + t;
    v[g] = i < B ? place_key(iters[i], support, i) : -1;
  }
#pragma unroll
  for (int k = 0; k < kPer; k++)
    if (t + k * kPlaceThreads < kN) cnt[t + k * kPlaceThreads] = 0;
  __shared__ uint32_t tot[kPlaceKeys], before[kPlaceKeys];
  if (gridDim.x > 1) {
    // several workgroups: robots of the harder bins anywhere, and of my bin in the workgroups before mine
    if (t < kPlaceKeys) { tot[t] = 0; before[t] = 0; }
    __syncthreads();
    for (unsigned e = t; e < gridDim.x * kPlaceKeys; e += kPlaceThreads) {
      const unsigned blk = e / kPlaceKeys, b = e - blk * kPlaceKeys;
      const uint32_t c = blockhist[e];
      atomicAdd(&tot[b], c);
      if (blk < blockIdx.x) atomicAdd(&before[b], c);
    }
    __syncthreads();
    if (t < kPlaceKeys) {
      uint32_t harder = 0;
      for (int b = 0; b < kPlaceKeys; b++) harder += b < t ? tot[b] : 0u;
      bin_base[t] = harder + before[t];
    }
  }
  __syncthreads();
  uint32_t ord[G];
#pragma unroll
  for (int g = 0; g < G; g++) ord[g] = v[g] >= 0 ? atomicAdd(&cnt[(v[g] * G + g) * kPlaceWaves + wave], 1u) : 0u;
  __syncthreads();
  // exclusive scan of the counters in place: kPer consecutive counters per lane, then lanes, then wavefronts
  uint32_t c[kPer], mine = 0;
#pragma unroll
  for (int k = 0; k < kPer; k++) { c[k] = t * kPer + k < kN ? cnt[t * kPer + k] : 0u; mine += c[k]; }
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    incl += lane >= d ? o : 0u;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t run = incl - mine;
#pragma unroll
  for (int w = 0; w < kPlaceWaves; w++) run += w < wave ? wtot[w] : 0u;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    if (t * kPer + k < kN) cnt[t * kPer + k] = run;
    run += c[k];
  }
  __syncthreads();
  const int64_t W = (B + 3) / 4;
#pragma unroll
  for (int g = 0; g < G; g++) {
    if (v[g] < 0) continue;
    int64_t r = cnt[(v[g] * G + g) * kPlaceWaves + wave] + ord[g];
    if (gridDim.x > 1) r += (int64_t)bin_base[v[g]] - cnt[v[g] * G * kPlaceWaves]; // my workgroup's harder bins come off
    int64_t slot = r;
    if (!throughput) {
      const int64_t e = B - 1 - r;
      slot = r < W ? 4 * r : 4 * (e / 3) + 1 + e % 3;
    }
    order[slot] = (int32_t)(first + g * kPlaceThreads + t);
  }
}

// The same placement inside a placed launch, for the extra workgroups it carries when the caller asks for the next launch's
// placement (qlamd_placement::next_robot_order): they run in the shadow of the solve -- a launch of a few thousand robots
// lasts 9-24 us, one of these wavefronts a few us for its 1024 robots -- instead of as launches of their own between two
// control steps (6-9 us).  One wavefront (= workgroup) per `chunk` robots, the first workgroups of the grid:
// rounds of 64 robots; counters [key][round] in the workgroup's LDS (the solve's 13.5 KB).  Pass 1 counts (LDS atomics
// without a return value), an exclusive scan in (key, round) order turns the counters into first ranks within the chunk,
// pass 2 takes each robot's rank as the return value of one more atomic on its counter (lanes that meet on a counter are
// served in lane order: rank order = index order within a key).  With several chunks a wavefront needs the others' counts
// per key between the scan and pass 2: each leaves its own in the context's scratch and they meet at a barrier in global
// memory (arrivals + generation, agent scope; the workgroups in front of a grid are dispatched first and all at once, so
// they can wait for each other; the barrier resets itself, so a hipGraph can replay the launch).  Keys: the iteration
// count's bin, and with a sorted placement also the robot's class (place_key: robots on more than two legs first).
// A lone wavefront issues one instruction per ~5.5 cycles and waits out every memory round trip, so the loop bodies are
// counted in instructions (32-bit index arithmetic, the division by 3 as a multiplication, one clamp per count),
// kShadowChunk rounds have their loads in flight together, and the last, ragged rounds are the only ones that check
// indices.  Measured and dropped (profiles/r5/placed_probe.txt): ONE atomic pass whose return values are kept until the
// scan is done -- in registers (13 000 instructions of unrolled code), as bytes in LDS (16 us at 4096 robots: sub-word LDS
// stores) or as words of four rounds (14.6 us, and 16 KB of LDS per workgroup cost every placed launch 0.3 us).
constexpr int kShadowLdsBytes = (4 * kTabPerLeg + 4 * coop::kCoopLdsDoubles + coop::kCoopNrmDoubles) * 8;
// robots per shadow wavefront.  Below the throughput form's batches: 1024 with a warm start (4096 robots: 4 wavefronts, 6 us
// -- shorter than the shortest solve, a warm-started calm batch's 9.5 us), 2048 without (8 us against the 13.4 us of a calm
// batch solved cold: every shadow wavefront shares a SIMD with a wavefront that solves, and two of them cost the placed
// loop of 4096 robots 0.3 us less than four: profiles/r5/ab_shadow_blocks.txt).  4096 from there (65 536 robots: 16
// wavefronts, 13 us of a 70 us launch; with 64 of them every one reads 64 x 48 counts and the launch is 3-5 us longer).
constexpr int kShadowChunkWarm = 1024, kShadowChunkCold = 2048, kShadowChunkLarge = 4096, kShadowMaxBlocks = 256;
static_assert((kShadowChunkLarge / 64) * kPlaceKeys * 4 + kPlaceKeys * 4 <= kShadowLdsBytes, "the shadow wavefront's counters live in the solve's LDS");
constexpr int kShadowChunk = 16; // rounds whose loads are in flight together: a chunk's 1024 robots
typedef __attribute__((address_space(3))) uint32_t lds_u32; // (a generic pointer would turn the atomics into flat ones)
template <bool kThroughput>
__device__ __forceinline__ uint32_t place_slot(uint32_t rk, uint32_t B, uint32_t W) {
  if constexpr (kThroughput) return rk;
  const uint32_t e = B - 1u - rk, q = __umulhi(e, 0xAAAAAAABu) >> 1; // e / 3
  return rk < W ? 4u * rk : 4u * q + 1u + (e - 3u * q);
}
// sidx of S: my chunk; support: the robots' support flags when the keys carry the class (sorted placement), else NULL
__device__ __forceinline__ void placement_wave(const int32_t *__restrict__ iters, int64_t B64, int throughput,
                                               int32_t *__restrict__ order, lds_u32 *cnt, uint32_t sidx, uint32_t S,
                                               uint32_t chunk, const uint32_t *__restrict__ support, uint32_t *__restrict__ ghist,
                                               uint32_t *__restrict__ gsync, uint32_t max_polls) {
  const uint32_t lane = threadIdx.x & 63u, B = (uint32_t)B64, W = (B + 3u) >> 2;
  const uint32_t lo = sidx * chunk;
  const uint32_t n = lo >= B ? 0u : (B - lo < chunk ? B - lo : chunk); // my robots
  const uint32_t R = (n + 63u) >> 6, full = n >> 6;      // rounds, rounds without a missing robot
  const bool classes = throughput && support != nullptr;
  const uint32_t nkeys = classes ? kPlaceKeys : kPlaceBins, N = nkeys * R; // counters
  lds_u32 *adj = cnt + N;                                // what turns a rank within the chunk into the rank of the batch, per key
  for (uint32_t k = lane; k < N + nkeys; k += 64) cnt[k] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const auto run = [&](auto Classes, auto Thr) {
    constexpr bool kClasses = decltype(Classes)::value, kThr = decltype(Thr)::value;
    const auto key_of = [&](int v, uint32_t w) -> uint32_t {
      uint32_t key = (uint32_t)place_bin(v);
      if constexpr (kClasses) {
        const int legs = ((w & 0xFFu) ? 1 : 0) + ((w & 0xFF00u) ? 1 : 0) + ((w & 0xFF0000u) ? 1 : 0) + ((w & 0xFF000000u) ? 1 : 0);
        key += legs <= 2 ? (uint32_t)kPlaceBins : 0u;
      }
      return key;
    };
    // one pass over my robots: kPass 1 counts, kPass 2 ranks and stores
    const auto pass = [&](auto Pass) {
      constexpr int kPass = decltype(Pass)::value;
      for (uint32_t r0 = 0; r0 < R; r0 += kShadowChunk) {
        if (r0 + kShadowChunk <= full) {
          int v[kShadowChunk];
          uint32_t w[kShadowChunk], rk[kShadowChunk];
#pragma unroll
          for (int k = 0; k < kShadowChunk; k++) {
            v[k] = iters[lo + (r0 + k) * 64u + lane];
            w[k] = kClasses ? support[lo + (r0 + k) * 64u + lane] : 0u;
          }
#pragma unroll
          for (int k = 0; k < kShadowChunk; k++) {
            const uint32_t key = key_of(v[k], w[k]);
            w[k] = key;
            lds_u32 *c = cnt + key * R + (r0 + k);
            if constexpr (kPass == 1) (void)__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else rk[k] = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if constexpr (kPass == 2) {
#pragma unroll
            for (int k = 0; k < kShadowChunk; k++)
              order[place_slot<kThr>(rk[k] + adj[w[k]], B, W)] = (int32_t)(lo + (r0 + k) * 64u + lane);
          }
        } else {
          for (uint32_t r = r0; r < R && r < r0 + kShadowChunk; r++) {
            const uint32_t i = r * 64u + lane;
            if (i < n) {
              const uint32_t key = key_of(iters[lo + i], kClasses ? support[lo + i] : 0u);
              lds_u32 *c = cnt + key * R + r;
              if constexpr (kPass == 1) (void)__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              else {
                const uint32_t rk = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                order[place_slot<kThr>(rk + adj[key], B, W)] = (int32_t)(lo + i);
              }
            }
          }
        }
      }
    };
    pass(std::integral_constant<int, 1>{});
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- exclusive scan in (key, round) order: `per` consecutive counters per lane
    {
      const uint32_t per = (N + 63u) >> 6, k0 = lane * per, k1 = k0 + per < N ? k0 + per : N;
      uint32_t mine = 0;
      for (uint32_t k = k0; k < k1; k++) mine += cnt[k];
      uint32_t incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        incl += lane >= (uint32_t)d ? o : 0u;
      }
      uint32_t running = incl - mine;
      for (uint32_t k = k0; k < k1; k++) { const uint32_t c = cnt[k]; cnt[k] = running; running += c; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (S > 1) {
      // ---- the other chunks: lane k < nkeys owns key k.  My count of it goes to the scratch, the barrier, then
      //      rank of the batch = robots of harder keys anywhere + robots of my key in the chunks before mine + rank in my chunk
      const uint32_t start = (lane < nkeys && R > 0) ? cnt[lane * R] : 0u;
      const uint32_t next = (lane + 1u < nkeys && R > 0) ? cnt[(lane + 1u) * R] : n;
      if (lane < nkeys) __hip_atomic_store(ghist + sidx * kPlaceKeys + lane, R > 0 ? next - start : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      // The barrier: arrivals in gsync[0], its state in gsync[1].  A launch finds the state at an even value g and leaves it at
      // g + 2 (everybody arrived: the last one says so) or at g + 4 (somebody gave up waiting -- g + 1, odd -- and the last one
      // to arrive closed the launch).  Both transitions away from g are compare-and-swaps, so a launch has ONE outcome, and
      // every wavefront can tell which whenever it looks (odd: given up; g + 2: complete; g + 4: given up and closed).  The
      // waiting is bounded (a second or two: a device shared with something that keeps these wavefronts from running together
      // must not hang), and a launch that gives up writes the identity order -- a valid placement, which costs the next launch
      // time, never a result.
      int gave_up = 0;
      if (lane == 0) {
        const uint32_t g = __hip_atomic_load(gsync + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t arrived = __hip_atomic_fetch_add(gsync, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = arrived == S - 1u;
        if (last) __hip_atomic_store(gsync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g & 1u) { // somebody gave up before I got here
          gave_up = 1;
          if (last) __hip_atomic_store(gsync + 1, g + 3u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (last) {
          uint32_t now = g;
          if (!__hip_atomic_compare_exchange_strong(gsync + 1, &now, g + 2u, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(gsync + 1, g + 4u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = 1;
          }
        } else {
          uint32_t now = g;
          for (unsigned spin = 0;; spin++) {
            now = __hip_atomic_load(gsync + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (now != g) break;
            if (spin >= max_polls) {
              if (__hip_atomic_compare_exchange_strong(gsync + 1, &now, g + 1u, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) now = g + 1u;
              break; // (lost the exchange: `now` holds what the winner wrote)
            }
            __builtin_amdgcn_s_sleep(4);
          }
          gave_up = now != g + 2u ? 1 : 0;
        }
      }
      gave_up = __builtin_amdgcn_readfirstlane(gave_up);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (gave_up) {
        for (uint32_t i = lane; i < n; i += 64) order[lo + i] = (int32_t)(lo + i);
        if (sidx == 0 && lane == 0) atomicAdd(gsync + kSyncGiveUps, 1u); // one count per launch (QLAMD_COUNTER_PLACEMENT_GIVE_UPS)
        return;
      }
      uint32_t total = 0, before = 0;
      if (lane < nkeys) {
        for (uint32_t b = 0; b < S; b++) {
          const uint32_t h = __hip_atomic_load(ghist + b * kPlaceKeys + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          total += h;
          before += b < sidx ? h : 0u;
        }
      }
      uint32_t incl = total;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        incl += lane >= (uint32_t)d ? o : 0u;
      }
      if (lane < nkeys) adj[lane] = (incl - total) + before - start;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    pass(std::integral_constant<int, 2>{});
  };
  if (classes) run(std::true_type{}, std::true_type{});
  else if (throughput) run(std::false_type{}, std::true_type{});
  else run(std::false_type{}, std::false_type{});
}

// The second attempt of the rows whose warm start was rejected (balance_coop_kernel<..., kWarm = true>, below): the plain kernel's
// body as a function of its own that ENDS THE WAVEFRONT -- nothing of the caller's is live across the call -- and takes what it
// needs from the kernel's argument segment again (coop::kernel_arguments_again: the caller passes the segment's address -- a
// function that is not a kernel has none of its own -- and keeps nothing else for it).
struct BalanceCoopArgs { const DeviceParams *Pp; StatePtrs s; int64_t B; double *tau, *grf; int32_t *status; }; // the kernel's parameters
template <bool kPerLeg, int kMinWaves>
__device__ __attribute__((noinline, noreturn)) void balance_cold_retry(const BalanceCoopArgs *args, double *lds, bool rejected) {
  const BalanceCoopArgs &a = *args;
  double *tab = lds, *rows = lds + 4 * kTabPerLeg, *nrm = rows + 4 * coop::kCoopLdsDoubles;
  const int row = threadIdx.x >> 4;
  const unsigned blk = blockIdx.x - (unsigned)a.s.shadow_blocks;
  int64_t ir = (int64_t)blk * 4 + row;
  const bool inside = ir < a.B;
  if (!inside) ir = a.B - 1;
  if (a.s.order) {
    const int64_t o = a.s.order[ir];
    ir = (inside && o >= 0 && o < a.B) ? o : a.B - 1;
  }
  const coop::CoopPtrs cold{a.s.q, a.s.pos, a.s.quat, a.s.linvel, a.s.angvel, a.s.dpos, a.s.dquat, a.s.dlinvel, a.s.dangvel, a.s.stance,
                            a.s.normals, a.s.wrench, a.s.live, a.s.support_only, a.s.iterations, nullptr, nullptr, nullptr, a.s.record_doubles};
  (void)coop::coop_robot<kPerLeg, 64, false, kMinWaves == 3>(*a.Pp, cold, ir, rejected, tab, rows + row * coop::kCoopLdsDoubles, nrm, a.tau, a.grf, a.status);
  if (rejected && (threadIdx.x & 15) == 0 && a.s.working_set) a.s.working_set[ir] = 0u;
  __builtin_amdgcn_endpgm();
}

// Latency form: 16 lanes per robot, 4 robots per wavefront (balance_coop.hpp), kCoopWaves wavefronts per workgroup.
// One wavefront per workgroup is the measured optimum: four (one workgroup per compute unit, the model table in LDS
// shared) costs 2.3 us at 4096 robots and 20 % at 65536 -- the barrier behind the table ties the start of four
// wavefronts together and workgroups leave their compute unit only when their slowest wavefront has finished.
constexpr int kCoopWaves = 1;
#ifndef QLAMD_THROUGHPUT_BATCH
#define QLAMD_THROUGHPUT_BATCH 16384 // robots from which the three-wavefront form of the balance kernel runs
#endif
#ifndef QLAMD_THREE_WAVE_WARM_BATCH
// ... of the warm-started kernel: later, because its two-wavefront form installs by rounds and lets robots without a set build
// one by rounds, which the 168 registers of the other form have no room for (trot, placed + warm loop, us per step, two- against
// three-wavefront form: 16 384 robots 30.0 / 32.6, 20 480: 33.6 / 34.3, 24 576: 37.9 / 37.1, 32 768: 45.6 / 42.8, 65 536:
// 76.7 / 70.5 -- profiles/r6/ab_three_wave_threshold.txt)
#define QLAMD_THREE_WAVE_WARM_BATCH 22528
#endif
// kMinWaves wavefronts per SIMD at least: 2 (at most 256 registers: the large-batch throughput halves without it) for the
// latency form, 3 (at most 168 registers; 12 wavefronts x 13 056 bytes of LDS fit a compute unit) for the throughput form
// that large batches take (QLAMD_THROUGHPUT_BATCH): two wavefronts of dependent instruction streams cannot fill a SIMD's
// issue port (2 x one instruction per 5.5 cycles against one per 4), three can.
// kPlaced: the caller says which robot sits in which slot of the launch (qlamd_balance_solve_placed_batch): the four robots
// of a wavefront run in lockstep, so a wavefront lasts as long as the union of their passes, and a launch of a few
// thousand robots as long as its slowest wavefront -- WHO shares a wavefront decides both.  The slot's
// robot index is one more load in front of the robot's own (a dependent round trip at the head of the launch), which is
// why the plain entry keeps a kernel without it.  An entry outside [0, B) leaves its row empty.
// kWarm (with kPlaced): every robot's QP starts from the working set the caller hands in (force_qp_coop.hpp).
template <bool kPerLeg, int kMinWaves, bool kPlaced = false, bool kWarm = false>
__global__ __launch_bounds__(64 * kCoopWaves, kMinWaves) void balance_coop_kernel(const DeviceParams *__restrict__ Pp, const StatePtrs s,
                                                                      int64_t B, double *__restrict__ tau,
                                                                      double *__restrict__ grf, int32_t *__restrict__ status) {
  __shared__ double lds[4 * kTabPerLeg + 4 * kCoopWaves * coop::kCoopLdsDoubles + kCoopWaves * coop::kCoopNrmDoubles];
  static_assert(sizeof(lds) >= kShadowLdsBytes || kCoopWaves != 1, "the shadow wavefronts' counters live in the solve's LDS");
  double *tab = lds, *rows = lds + 4 * kTabPerLeg, *nrm = rows + 4 * kCoopWaves * coop::kCoopLdsDoubles;
  const DeviceParams &P = *Pp;
  const int row = threadIdx.x >> 4, wave = threadIdx.x >> 6;
  unsigned block = blockIdx.x;
  QL_BLOCK_STAMP(0);
  if constexpr (kPlaced) {
    // the first workgroups of a launch that also places the next one (they start first and have the whole launch to finish in)
    if (s.shadow_blocks) {
      if (block < (unsigned)s.shadow_blocks) {
        placement_wave(s.prev_iterations, B, s.place_throughput, s.next_order, (lds_u32 *)lds, block, (uint32_t)s.shadow_blocks,
                       (uint32_t)s.shadow_chunk, s.place_throughput ? reinterpret_cast<const uint32_t *>(s.stance) : nullptr, s.place_hist, s.place_sync,
                       s.place_wait);
        return;
      }
      block -= (unsigned)s.shadow_blocks;
      // the wavefronts that solve go first wherever one of them shares a SIMD with a shadow wavefront (which has the whole
      // launch to finish in): without it the placed loop of 4096 robots is 0.4 us longer with four shadow wavefronts than
      // with one (profiles/r5/ab_shadow_blocks.txt)
      __builtin_amdgcn_s_setprio(3);
    }
  }
  int64_t i = (int64_t)block * (4 * kCoopWaves) + row;
  bool live = i < B;
  if (!live) i = B - 1;
  if constexpr (kPlaced) {
    if constexpr (kMinWaves == 2) { // (QLAMD_PLACEMENT_NONE: the batch order; the 168-register form has no register for it -- a launch of its own there)
      if (s.shadow_blocks == 0 && s.next_order && live && (threadIdx.x & 15) == 0) s.next_order[i] = (int32_t)i;
    }
    if (s.order) {
      const int64_t o = s.order[i];
      live = live && o >= 0 && o < B;
      i = live ? o : B - 1;
    }
  }
  const coop::CoopPtrs cp{s.q, s.pos, s.quat, s.linvel, s.angvel, s.dpos, s.dquat, s.dlinvel, s.dangvel, s.stance,
                          s.normals, s.wrench, s.live, s.support_only, kPlaced ? s.iterations : nullptr,
                          kWarm ? s.prev_working_set : nullptr, kWarm ? s.working_set : nullptr, kWarm ? s.warm_retries : nullptr,
                          s.record_doubles};
#ifdef QLAMD_STAMPS
#pragma unroll 1
  for (int rep = 0; rep < 2; rep++) // second pass runs with a warm instruction cache
#endif
  // (inputs parked in LDS across the first form of the QP: the 168-register form solving cold -- no scratch then, 1-2 % on 65 536
  // to a million robots; the warm-started kernel is 3 % faster with them in registers and 20 bytes of scratch around the loop)
  {
    const bool rejected = coop::coop_robot<kPerLeg, 64 * kCoopWaves, kWarm, kMinWaves == 3 && !kWarm && kPlaced, kMinWaves == 3, kPlaced>(
        P, cp, i, live, tab, rows + row * coop::kCoopLdsDoubles, nrm + wave * coop::kCoopNrmDoubles, tau, grf, status);
    QL_BLOCK_STAMP(3);
    if constexpr (kWarm) {
      // A warm start must never cost an answer (balance_coop.hpp): the rows whose warm start was rejected are solved again, cold,
      // by this wavefront -- the plain kernel's body behind a scalar branch, everything it needs fetched again from the argument
      // segment (coop::kernel_arguments_again), the other rows riding along empty; the robot's working set comes back 0.
      if (__builtin_expect(P.warm_fallback && __builtin_amdgcn_ballot_w64(rejected) != 0ull, 0)) {
        __syncthreads(); // (one wavefront: the first attempt's LDS reads are done before the table is staged again)
        balance_cold_retry<kPerLeg, kMinWaves>(coop::kernel_arguments_again<BalanceCoopArgs>(), lds, rejected);
      }
    }
  }
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

} // namespace

extern "C" {

void qlamd_balance_default_params(qlamd_balance_params *p) { if (p) default_balance_params(p); }
void qlamd_default_robot_model(qlamd_robot_model *m) { if (m) default_robot_model(m); }
int qlamd_version(void) { return QLAMD_VERSION_MAJOR * 1000 + QLAMD_VERSION_MINOR; }


const char *qlamd_strerror(int code) {
  switch (code) {
    case QLAMD_OK: return "ok";
    case QLAMD_ERR_INVALID_ARGUMENT: return "invalid argument";
    case QLAMD_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case QLAMD_ERR_HIP: return "HIP runtime error";
    case QLAMD_ERR_NOT_LOADED: return "parameters not loaded";
    case QLAMD_ERR_OUT_OF_MEMORY: return "out of device memory";
    case QLAMD_ERR_BUSY: return "another thread is inside a call on this context";
    case QLAMD_ERR_NEEDS_RESERVE: return "device scratch would have to grow inside a stream capture: call qlamd_reserve first";
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
  ctx->place_ws = nullptr;
  ctx->place_ws_bytes = 0;
  ctx->place_sync = nullptr;
  ctx->placement_wait = 1u << 13;
  ctx->has_next_placement = false;
  ctx->tick_place_state = nullptr;
  ctx->tick_place_batch = 0;
  ctx->tick_place_count = 0;
  ctx->on_failure = QLAMD_ON_FAILURE_ZERO;
  ctx->dynamics_form = QLAMD_DYNAMICS_AUTO;
  ctx->state_record_doubles = 0;
  ctx->depth = 0;
  ctx->last_stream = nullptr;
  ctx->done_event = nullptr;
  ctx->done_recorded = false;
  ctx->had_work = false;
  ctx->multi_stream = false;
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
  // (the placement scratch for batches up to 1 M robots comes with the context: 48 KB, and no placement call of a
  // sensible size ever has to allocate inside a stream capture)
  const size_t place_bytes = (size_t)256 * kPlaceKeys * sizeof(uint32_t);
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&ctx->d_params, sizeof(DeviceParams)) != hipSuccess ||
      hipMemcpy(ctx->d_params, &ctx->params, sizeof(DeviceParams), hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(&ctx->place_ws, place_bytes) != hipSuccess || hipMalloc(&ctx->place_sync, 256) != hipSuccess ||
      hipMemset(ctx->place_sync, 0, 256) != hipSuccess) {
    if (ctx->d_params) (void)hipFree(ctx->d_params);
    if (ctx->place_ws) (void)hipFree(ctx->place_ws);
    if (ctx->place_sync) (void)hipFree(ctx->place_sync);
    delete ctx;
    return QLAMD_ERR_HIP;
  }
  ctx->place_ws_bytes = place_bytes;
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
  if (ctx->place_ws) (void)hipFree(ctx->place_ws);
  if (ctx->place_sync) (void)hipFree(ctx->place_sync);
  if (ctx->d_params) (void)hipFree(ctx->d_params);
  if (ctx->done_event) (void)hipEventDestroy(ctx->done_event);
  delete ctx;
}

int qlamd_set_robots_per_wave(qlamd_context *ctx, int rpw) {
  if (!ctx || !(rpw == 0 || rpw == 4 || rpw == 16 || rpw == 64)) return QLAMD_ERR_INVALID_ARGUMENT;
  ctx->rpw_override = rpw;
  return QLAMD_OK;
}

int qlamd_set_option(qlamd_context *ctx, int option, int value) {
  if (!ctx) return QLAMD_ERR_INVALID_ARGUMENT;
  QL_ENTER_NO_STREAM(ctx);
  switch (option) {
    case QLAMD_OPT_ON_FAILURE:
      if (value != QLAMD_ON_FAILURE_ZERO && value != QLAMD_ON_FAILURE_KEEP) return QLAMD_ERR_INVALID_ARGUMENT;
      ctx->on_failure = value;
      ctx->params.keep_on_failure = value == QLAMD_ON_FAILURE_KEEP;
      break;
    case QLAMD_OPT_REFINE_PASSES:
      if (value < 0 || value > 4) return QLAMD_ERR_INVALID_ARGUMENT;
      ctx->params.refine_passes = value;
      break;
    case QLAMD_OPT_DYNAMICS_FORM:
      if (value < QLAMD_DYNAMICS_AUTO || value > QLAMD_DYNAMICS_ROW) return QLAMD_ERR_INVALID_ARGUMENT;
      ctx->dynamics_form = (int)value;
      return QLAMD_OK;
    case QLAMD_OPT_PLACEMENT_WAIT:
      if (value < 0) return QLAMD_ERR_INVALID_ARGUMENT;
      ctx->placement_wait = (unsigned)value;
      return QLAMD_OK;
    case QLAMD_OPT_STATE_LAYOUT:
      if (value != QLAMD_STATE_FIELDS && value != QLAMD_STATE_RECORDS) return QLAMD_ERR_INVALID_ARGUMENT;
      ctx->state_record_doubles = value == QLAMD_STATE_RECORDS ? QLAMD_STATE_RECORD_DOUBLES : 0;
      return QLAMD_OK;
    case QLAMD_OPT_WARM_FALLBACK:
      if (value < 0 || value > 2) return QLAMD_ERR_INVALID_ARGUMENT;
      ctx->params.warm_fallback = value;
      break;
    default: return QLAMD_ERR_INVALID_ARGUMENT;
  }
  // the device copy of the parameters: after everything already queued on the context has read the old one
  if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(ctx->d_params, &ctx->params, sizeof(DeviceParams), hipMemcpyHostToDevice) != hipSuccess)
    return QLAMD_ERR_HIP;
  return QLAMD_OK;
}

int qlamd_get_counter(qlamd_context *ctx, int counter, int64_t *value) {
  if (!ctx || !value || (counter != QLAMD_COUNTER_PLACEMENT_GIVE_UPS && counter != QLAMD_COUNTER_WARM_RETRIES)) return QLAMD_ERR_INVALID_ARGUMENT;
  QL_ENTER_NO_STREAM(ctx);
  uint32_t v = 0;
  const uint32_t *src = (const uint32_t *)ctx->place_sync + (counter == QLAMD_COUNTER_PLACEMENT_GIVE_UPS ? kSyncGiveUps : kSyncWarmRetries);
  if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(&v, src, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
    return QLAMD_ERR_HIP;
  *value = (int64_t)v;
  return QLAMD_OK;
}

} // extern "C"

// live: device pointer [B] or NULL (whole tick, QLAMD_MEM_DEVICE only): robots with 0 are left alone
namespace {
// qlamd_placement_from_iterations on device pointers (the caller holds the context's guard)
// d_support: the robots' support flags ([B][4] bytes, device) or NULL -- the class of the sorted placement (place_key)
int launch_placement(qlamd_context *ctx, const int32_t *d_it, int64_t batch, int throughput, int32_t *d_ord, hipStream_t st,
                     const uint8_t *d_support = nullptr) {
  const uint32_t *sup = throughput ? reinterpret_cast<const uint32_t *>(d_support) : nullptr;
  if (batch <= kPlaceRounds * kPlaceThreads)
    hipLaunchKernelGGL(placement_kernel<kPlaceRounds>, dim3(1), dim3(kPlaceThreads), 0, st, d_it, batch, throughput, sup, nullptr, d_ord);
  else {
    // several workgroups (the LDS atomics of one compute unit serve about one robot per cycle: 4096 robots per workgroup
    // keep a launch at 3-4 us whatever the batch): their counts per bin go through the context's placement scratch
    // (sized for 1 M robots when the context is created; growing it is an allocation, which a stream capture cannot take)
    const int64_t per_block = (int64_t)kPlaceRounds * kPlaceThreads;
    const unsigned nb = (unsigned)((batch + per_block - 1) / per_block);
    const size_t need = (size_t)nb * kPlaceKeys * sizeof(uint32_t);
    if (ctx->place_ws_bytes < need) {
      if (CallGuard::capturing(st)) return QLAMD_ERR_NEEDS_RESERVE;
      if (hipStreamSynchronize(st) != hipSuccess) return QLAMD_ERR_HIP;
      if (ctx->place_ws) (void)hipFree(ctx->place_ws);
      ctx->place_ws = nullptr;
      ctx->place_ws_bytes = 0;
      if (hipMalloc(&ctx->place_ws, need) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
      ctx->place_ws_bytes = need;
    }
    hipLaunchKernelGGL(placement_hist_kernel, dim3(nb), dim3(kPlaceThreads), 0, st, d_it, batch, per_block, sup, (uint32_t *)ctx->place_ws);
    hipLaunchKernelGGL(placement_kernel<kPlaceRounds>, dim3(nb), dim3(kPlaceThreads), 0, st, d_it, batch, throughput, sup,
                       (const uint32_t *)ctx->place_ws, d_ord);
  }
  return hipGetLastError() == hipSuccess ? QLAMD_OK : QLAMD_ERR_HIP;
}
bool throughput_policy(int policy, int64_t batch) {
  return policy == QLAMD_PLACEMENT_THROUGHPUT || (policy == QLAMD_PLACEMENT_AUTO && batch >= QLAMD_THROUGHPUT_BATCH);
}
bool valid_policy(int policy) {
  return policy == QLAMD_PLACEMENT_AUTO || policy == QLAMD_PLACEMENT_LATENCY || policy == QLAMD_PLACEMENT_THROUGHPUT ||
         policy == QLAMD_PLACEMENT_NONE;
}
// What QLAMD_PLACEMENT_AUTO means for a call.  Without a warm start: the latency policy below QLAMD_THROUGHPUT_BATCH robots, the
// throughput policy from there.  With one (measured on trajectories, profiles/r6/ab_warm_policies.txt: us per step of the loop,
// latency / throughput / no placement): 4096 robots static 14.7 / 14.7 / 14.4, trot 19.9 / 19.7 / 19.6; 8192: 20.8 / 20.0 / 19.1
// and 25.5 / 23.3 / 23.8; 12 288: 25.7 / 25.3 / 25.0 and 31.5 / 26.5 / 29.1 -- a warm-started launch lasts as long as its
// slowest ROBOT (alone_probe.txt), so who shares a wavefront with whom no longer matters while every wavefront has a SIMD to
// itself, and the latency policy, which starts the hard robots together, puts two slow wavefronts on one SIMD as soon as there
// are two a SIMD: no placement up to QLAMD_WARM_UNPLACED_BATCH robots, the throughput policy (with its support classes) above.
#ifndef QLAMD_WARM_UNPLACED_BATCH
#define QLAMD_WARM_UNPLACED_BATCH 4096
#endif
int effective_policy(int policy, int64_t batch, bool warm) {
  if (policy != QLAMD_PLACEMENT_AUTO) return policy;
  if (warm) return batch <= QLAMD_WARM_UNPLACED_BATCH ? QLAMD_PLACEMENT_NONE : QLAMD_PLACEMENT_THROUGHPUT;
  return batch >= QLAMD_THROUGHPUT_BATCH ? QLAMD_PLACEMENT_THROUGHPUT : QLAMD_PLACEMENT_LATENCY;
}
__global__ void identity_order_kernel(int32_t *order, int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < B) order[i] = (int32_t)i;
}
} // namespace

int qlamd::rt::placement_launch(qlamd_context *ctx, const int32_t *d_iterations, int64_t batch, int policy, int32_t *d_order,
                                hipStream_t st) {
  if (policy == QLAMD_PLACEMENT_NONE) { // the batch order
    hipLaunchKernelGGL(identity_order_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, d_order, batch);
    return hipGetLastError() == hipSuccess ? QLAMD_OK : QLAMD_ERR_HIP;
  }
  return launch_placement(ctx, d_iterations, batch, throughput_policy(policy, batch) ? 1 : 0, d_order, st);
}

// pl: the placed entries' arrays (NULL otherwise), in the memory space of the call
int qlamd::rt::balance_impl(qlamd_context *ctx, const qlamd_state_batch *in_user, const double *wrench, const uint8_t *live, int support_only,
                            int64_t batch, double *joint_effort, double *contact_force, int32_t *status, int memory,
                            void *stream, const qlamd_placement *pl) {
  if (!ctx || !in_user || batch < 0 || !joint_effort || !status) return QLAMD_ERR_INVALID_ARGUMENT;
  const int32_t *order = pl ? pl->robot_order : nullptr;
  int32_t *iterations = pl ? pl->iterations : nullptr;
  const int32_t *prev_iterations = pl ? pl->prev_iterations : nullptr;
  int32_t *next_order = pl ? pl->next_robot_order : nullptr;
  if ((prev_iterations != nullptr) != (next_order != nullptr)) return QLAMD_ERR_INVALID_ARGUMENT;
  if (next_order && !valid_policy(pl->policy)) return QLAMD_ERR_INVALID_ARGUMENT;
  if (next_order && (next_order == order || prev_iterations == iterations)) return QLAMD_ERR_INVALID_ARGUMENT; // read and written by one launch
  const uint32_t *prev_ws = pl ? pl->prev_working_set : nullptr;
  uint32_t *ws = pl ? pl->working_set : nullptr;
  const bool warm = prev_ws || ws;
  const int policy = next_order ? effective_policy(pl->policy, batch, warm) : QLAMD_PLACEMENT_NONE;
  if (warm && memory != QLAMD_MEM_DEVICE) return QLAMD_ERR_INVALID_ARGUMENT; // (a host-buffer call is bound by its copies)
  // (prev_ws == ws is fine: a robot's set is read and written by its own 16 lanes only -- updated in place)
  const bool placed = order || iterations || next_order || warm;
  // the one-lane kernels of qlamd_set_robots_per_wave know no placement (a lane is a robot there: nothing is shared)
  if (placed && pick_rpw(ctx, batch) != 4) return QLAMD_ERR_INVALID_ARGUMENT;
  if (placed && batch > INT32_MAX) return QLAMD_ERR_INVALID_ARGUMENT;
  if (order && memory == QLAMD_MEM_HOST) {
    // host memory can be checked: every robot exactly once
    std::vector<uint8_t> seen((size_t)batch, 0);
    for (int64_t k = 0; k < batch; k++) {
      const int32_t o = order[k];
      if (o < 0 || o >= batch || seen[(size_t)o]) return QLAMD_ERR_INVALID_ARGUMENT;
      seen[(size_t)o] = 1;
    }
  }
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
  QL_ENTER(ctx, st);
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
    size_t off[17], total = 0;
    for (int k = 0; k < 12; k++) { off[k] = total; total += align256(sz[k]); }
    off[15] = total; total += align256(order ? B * 4 : 0);
    off[12] = total; total += align256(B * 96);
    off[13] = total; total += align256(B * 96);
    off[14] = total; total += align256(B * 4);
    off[16] = total; total += align256(iterations ? B * 4 : 0);
    int rc = ensure_ws(ctx, total);
    if (rc != QLAMD_OK) return rc;
    char *w = (char *)ctx->ws;
    small_host = total <= kSmallHostCall && ensure_pinned(ctx, kSmallHostCall) == QLAMD_OK;
    out_off = off[12];
    out_bytes = total - off[12];
    // QLAMD_ON_FAILURE_KEEP: a failed robot's entries are not written by the kernel, and the whole output region is
    // copied back below -- so the caller's efforts / forces go up with the inputs and come back untouched
    const bool keep = ctx->params.keep_on_failure != 0;
    const size_t up_bytes = keep ? off[14] : off[12];
    if (small_host) {
      // a single robot or a few: a dozen separate pageable copies cost ~10 us each; pack, copy once
      char *h = (char *)ctx->pinned;
      for (int k = 0; k < 12; k++)
        if (sz[k]) memcpy(h + off[k], src[k], sz[k]);
      if (order) memcpy(h + off[15], order, B * 4);
      if (keep) {
        memcpy(h + off[12], joint_effort, B * 96);
        if (contact_force) memcpy(h + off[13], contact_force, B * 96);
      }
      if (hipMemcpyAsync(w, h, up_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return QLAMD_ERR_HIP;
    } else {
      for (int k = 0; k < 12; k++)
        if (sz[k] && hipMemcpyAsync(w + off[k], src[k], sz[k], hipMemcpyHostToDevice, st) != hipSuccess)
          return QLAMD_ERR_HIP;
      if (order && hipMemcpyAsync(w + off[15], order, B * 4, hipMemcpyHostToDevice, st) != hipSuccess) return QLAMD_ERR_HIP;
      if (keep) {
        if (hipMemcpyAsync(w + off[12], joint_effort, B * 96, hipMemcpyHostToDevice, st) != hipSuccess) return QLAMD_ERR_HIP;
        if (contact_force && hipMemcpyAsync(w + off[13], contact_force, B * 96, hipMemcpyHostToDevice, st) != hipSuccess)
          return QLAMD_ERR_HIP;
      }
    }
    s = StatePtrs{(const double *)(w + off[0]), (const double *)(w + off[1]), (const double *)(w + off[2]),
                  (const double *)(w + off[3]), (const double *)(w + off[4]), (const double *)(w + off[5]),
                  (const double *)(w + off[6]), (const double *)(w + off[7]), (const double *)(w + off[8]),
                  (const uint8_t *)(w + off[9]), in->surface_normal ? (const double *)(w + off[10]) : nullptr,
                  wrench ? (const double *)(w + off[11]) : nullptr, nullptr, 0,
                  order ? (const int32_t *)(w + off[15]) : nullptr, iterations ? (int32_t *)(w + off[16]) : nullptr,
                  nullptr, nullptr, 0};
    d_tau = (double *)(w + off[12]);
    d_grf = contact_force ? (double *)(w + off[13]) : nullptr;
    d_status = (int32_t *)(w + off[14]);
  } else {
    s = StatePtrs{in->joint_position, in->base_position, in->base_orientation, in->base_linear_velocity,
                  in->base_angular_velocity, in->desired_position, in->desired_orientation,
                  in->desired_linear_velocity, in->desired_angular_velocity, in->support_leg,
                  in->surface_normal, wrench, live, support_only, order, iterations, nullptr, nullptr, 0};
    // records instead of per-field arrays: the lane-cooperative kernels only (and never with an external wrench, whose entry
    // points the unused pose fields at joint_position)
    if (ctx->state_record_doubles && (pick_rpw(ctx, batch) != 4 || wrench)) return QLAMD_ERR_INVALID_ARGUMENT;
    s.record_doubles = ctx->state_record_doubles;
    s.prev_working_set = prev_ws;
    s.working_set = ws;
    s.warm_retries = (uint32_t *)ctx->place_sync + kSyncWarmRetries;
    // the next launch's placement: by extra wavefronts in front of this launch
    const int chunk = batch >= QLAMD_THROUGHPUT_BATCH ? kShadowChunkLarge : (warm ? kShadowChunkWarm : kShadowChunkCold);
    const int64_t shadows = (batch + chunk - 1) / chunk;
    const bool three_wave_form = !in->surface_normal && batch >= (warm ? QLAMD_THREE_WAVE_WARM_BATCH : QLAMD_THROUGHPUT_BATCH);
    if (next_order && policy == QLAMD_PLACEMENT_NONE) {
      if (!three_wave_form) s.next_order = next_order; // (shadow_blocks stays 0: written by the slots themselves)
    } else if (next_order && shadows <= kShadowMaxBlocks && pick_rpw(ctx, batch) == 4) {
      s.prev_iterations = prev_iterations;
      s.next_order = next_order;
      s.place_throughput = policy == QLAMD_PLACEMENT_THROUGHPUT ? 1 : 0;
      s.shadow_blocks = (int)shadows;
      s.shadow_chunk = chunk;
      s.place_hist = (uint32_t *)ctx->place_ws;
      s.place_sync = (uint32_t *)ctx->place_sync;
      s.place_wait = ctx->placement_wait;
    }
  }

  hipError_t e;
  switch (pick_rpw(ctx, batch)) {
    case 4: {
      const unsigned grid = (unsigned)((batch + 4 * kCoopWaves - 1) / (4 * kCoopWaves)) + (unsigned)s.shadow_blocks;
#define QL_LAUNCH_COOP(PERLEG, WAVES)                                                                                        \
  do {                                                                                                                       \
    if (warm)                                                                                                                \
      hipLaunchKernelGGL((balance_coop_kernel<PERLEG, WAVES, true, true>), dim3(grid), dim3(64 * kCoopWaves), 0, st,         \
                         ctx->d_params, s, batch, d_tau, d_grf, d_status);                                                   \
    else if (placed)                                                                                                         \
      hipLaunchKernelGGL((balance_coop_kernel<PERLEG, WAVES, true>), dim3(grid), dim3(64 * kCoopWaves), 0, st, ctx->d_params, \
                         s, batch, d_tau, d_grf, d_status);                                                                  \
    else                                                                                                                     \
      hipLaunchKernelGGL((balance_coop_kernel<PERLEG, WAVES, false>), dim3(grid), dim3(64 * kCoopWaves), 0, st,              \
                         ctx->d_params, s, batch, d_tau, d_grf, d_status);                                                   \
  } while (0)
      if (s.normals) QL_LAUNCH_COOP(true, 2);
      else if (batch >= (warm ? QLAMD_THREE_WAVE_WARM_BATCH : QLAMD_THROUGHPUT_BATCH)) QL_LAUNCH_COOP(false, 3);
      else QL_LAUNCH_COOP(false, 2);
#undef QL_LAUNCH_COOP
      e = hipGetLastError();
      break;
    }
    case 16: e = launch_balance<16>(ctx, s, batch, d_tau, d_grf, d_status, st); break;
    default: e = launch_balance<64>(ctx, s, batch, d_tau, d_grf, d_status, st); break;
  }
  if (e != hipSuccess) return QLAMD_ERR_HIP;
  if (next_order && memory == QLAMD_MEM_DEVICE && policy == QLAMD_PLACEMENT_NONE && !s.next_order) {
    hipLaunchKernelGGL(identity_order_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, next_order, batch);
    if (hipGetLastError() != hipSuccess) return QLAMD_ERR_HIP;
  }
  if (next_order && memory == QLAMD_MEM_DEVICE && !s.shadow_blocks && policy != QLAMD_PLACEMENT_NONE) { // more robots than the shadow wavefronts take: launches of their own
    const int rc = launch_placement(ctx, prev_iterations, batch, policy == QLAMD_PLACEMENT_THROUGHPUT ? 1 : 0, next_order, st, s.stance);
    if (rc != QLAMD_OK) return rc;
  }

  if (memory == QLAMD_MEM_HOST && small_host) {
    char *h = (char *)ctx->pinned, *w = (char *)ctx->ws;
    if (hipMemcpyAsync(h + out_off, w + out_off, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return QLAMD_ERR_HIP;
    memcpy(joint_effort, h + ((char *)d_tau - w), B * 96);
    if (contact_force) memcpy(contact_force, h + ((char *)d_grf - w), B * 96);
    memcpy(status, h + ((char *)d_status - w), B * 4);
    if (iterations) memcpy(iterations, h + ((char *)s.iterations - w), B * 4);
  } else if (memory == QLAMD_MEM_HOST) {
    if (hipMemcpyAsync(joint_effort, d_tau, B * 96, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (contact_force &&
        hipMemcpyAsync(contact_force, d_grf, B * 96, hipMemcpyDeviceToHost, st) != hipSuccess)
      return QLAMD_ERR_HIP;
    if (hipMemcpyAsync(status, d_status, B * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (iterations && hipMemcpyAsync(iterations, s.iterations, B * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return QLAMD_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return QLAMD_ERR_HIP;
  }
  if (next_order && memory == QLAMD_MEM_HOST)
    return qlamd_placement_from_iterations(ctx, prev_iterations, batch, policy, next_order, QLAMD_MEM_HOST, stream);
  return QLAMD_OK;
}

extern "C" {

int qlamd_balance_solve_batch(qlamd_context *ctx, const qlamd_state_batch *in, int64_t batch, double *joint_effort,
                              double *contact_force, int32_t *status, int memory, void *stream) {
  return balance_impl(ctx, in, nullptr, nullptr, 0, batch, joint_effort, contact_force, status, memory, stream);
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
  return balance_impl(ctx, &in, virtual_wrench, nullptr, 0, batch, joint_effort, contact_force, status, memory, stream);
}

int qlamd_balance_solve_placed_batch(qlamd_context *ctx, const qlamd_state_batch *in, int64_t batch, const qlamd_placement *placement,
                                     double *joint_effort, double *contact_force, int32_t *status, int memory, void *stream) {
  return balance_impl(ctx, in, nullptr, nullptr, 0, batch, joint_effort, contact_force, status, memory, stream, placement);
}

int qlamd_force_distribution_placed_batch(qlamd_context *ctx, const double *joint_position, const double *base_orientation,
                                          const uint8_t *support_leg, const double *surface_normal,
                                          const double *virtual_wrench, int64_t batch, const qlamd_placement *placement,
                                          double *joint_effort, double *contact_force, int32_t *status, int memory, void *stream) {
  if (!virtual_wrench) return QLAMD_ERR_INVALID_ARGUMENT;
  qlamd_state_batch in;
  memset(&in, 0, sizeof(in));
  in.joint_position = joint_position;
  in.base_orientation = base_orientation;
  in.support_leg = support_leg;
  in.surface_normal = surface_normal;
  return balance_impl(ctx, &in, virtual_wrench, nullptr, 0, batch, joint_effort, contact_force, status, memory, stream, placement);
}

int qlamd_place_next_call(qlamd_context *ctx, const qlamd_placement *placement) {
  if (!ctx) return QLAMD_ERR_INVALID_ARGUMENT;
  QL_ENTER_NO_STREAM(ctx);
  ctx->has_next_placement = false;
  if (!placement) return QLAMD_OK;
  if ((placement->prev_iterations != nullptr) != (placement->next_robot_order != nullptr)) return QLAMD_ERR_INVALID_ARGUMENT;
  if (placement->next_robot_order && !valid_policy(placement->policy)) return QLAMD_ERR_INVALID_ARGUMENT;
  if (placement->next_robot_order && (placement->next_robot_order == placement->robot_order || placement->prev_iterations == placement->iterations))
    return QLAMD_ERR_INVALID_ARGUMENT;
  ctx->next_placement = *placement;
  ctx->has_next_placement = true;
  return QLAMD_OK;
}

int qlamd_placement_from_iterations(qlamd_context *ctx, const int32_t *iterations, int64_t batch, int policy,
                                    int32_t *robot_order, int memory, void *stream) {
  if (!ctx || !iterations || !robot_order || batch < 0 || batch > INT32_MAX) return QLAMD_ERR_INVALID_ARGUMENT;
  if (!valid_policy(policy)) return QLAMD_ERR_INVALID_ARGUMENT;
  if (memory != QLAMD_MEM_DEVICE && memory != QLAMD_MEM_HOST) return QLAMD_ERR_INVALID_ARGUMENT;
  if (batch == 0) return QLAMD_OK;
  if (policy == QLAMD_PLACEMENT_NONE) { // the batch order
    if (memory == QLAMD_MEM_HOST) {
      for (int64_t i = 0; i < batch; i++) robot_order[i] = (int32_t)i;
      return QLAMD_OK;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
    QL_ENTER(ctx, (hipStream_t)stream);
    hipLaunchKernelGGL(identity_order_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, robot_order, batch);
    return hipGetLastError() == hipSuccess ? QLAMD_OK : QLAMD_ERR_HIP;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) return QLAMD_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  QL_ENTER(ctx, st);
  const size_t B = (size_t)batch;
  const int32_t *d_it = iterations;
  int32_t *d_ord = robot_order;
  Staged sg;
  if (memory == QLAMD_MEM_HOST) {
    const int a = sg.add(iterations, B * 4, true, false), o = sg.add(robot_order, B * 4, false, true);
    const int rc = sg.upload(ctx, st);
    if (rc != QLAMD_OK) return rc;
    d_it = sg.dev<const int32_t>(a);
    d_ord = sg.dev<int32_t>(o);
  }
  const int rc = launch_placement(ctx, d_it, batch, throughput_policy(policy, batch) ? 1 : 0, d_ord, st);
  if (rc != QLAMD_OK) return rc;
  return memory == QLAMD_MEM_HOST ? sg.finish(st) : QLAMD_OK;
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
  QL_ENTER(ctx, st);
  const size_t B = (size_t)batch;
  StatePtrs s{in->joint_position, in->base_position, in->base_orientation, in->base_linear_velocity,
              in->base_angular_velocity, in->desired_position, in->desired_orientation,
              in->desired_linear_velocity, in->desired_angular_velocity, in->support_leg, nullptr, nullptr, nullptr};
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
                  sg.dev<const double>(a8), nullptr, nullptr, nullptr, nullptr};
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
  QL_ENTER(ctx, st);
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

QLAMD_STAMPS_ACCESSOR(qlamd_debug_stamps)
QLAMD_BLOCK_STAMPS_ACCESSOR(qlamd_debug_block_stamps)
