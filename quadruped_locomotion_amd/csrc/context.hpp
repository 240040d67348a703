// What every translation unit of libqlamd.so shares on the host side: the context behind the opaque handle of
// include/qlamd.h, its device workspaces, the staging of host-buffer calls, and the LDS staging of the model table
// that the per-leg kernels use.
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "balance_core.hpp"
#include "qlamd.h"

struct qlamd_context {
  int device;
  qlamd::DeviceParams params;
  qlamd::DeviceParams *d_params; // device copy, read through scalar loads
  int rpw_override;
  int num_cu;
  double base_m, base_h[3], base_I[6]; // base_link about the base origin (whole-body entries)
  void *tick_ws;       // intermediates of qlamd_full_tick_batch (grown on demand)
  size_t tick_ws_bytes;
  void *place_ws;      // per-workgroup bin counts of qlamd_placement_from_iterations beyond 4096 robots
  size_t place_ws_bytes;
  void *place_sync;    // zero from the start.  Two words: the barrier of a placed launch's shadow wavefronts (balance_kernel.hip);
                       // two more: event counters (kSyncGiveUps, kSyncWarmRetries)
  qlamd_placement next_placement; // qlamd_place_next_call: taken (and cleared) by the next QP entry that knows placements
  bool has_next_placement;
  const void *tick_place_state;   // qlamd_tick_batch::placement_state of the last tick that had one, its batch and how many
  int64_t tick_place_batch;       // ticks of the placed loop have run on it
  int64_t tick_place_count;
  uint32_t *wire_tpl;  // two layout templates of robot_state_unpack_kernel (read one, write the other), or NULL
  int wire_flip;
  // HOST-memory mode staging (grown on demand)
  void *ws;
  size_t ws_bytes;
  void *pinned;        // page-locked mirror of the head of ws, for small host-buffer calls (one copy each way)
  size_t pinned_bytes;
  // options (qlamd_set_option): never read from the environment
  int on_failure, dynamics_form;
  int state_record_doubles;  // QLAMD_OPT_STATE_LAYOUT: 0 = one array per field, else the record length in doubles
  unsigned placement_wait;   // QLAMD_OPT_PLACEMENT_WAIT: polls the shadow wavefronts of a placed launch wait for each other
  // one call at a time (include/qlamd.h, "Threads and streams"): owner thread and nesting depth of the call in
  // progress, and the stream of the previous call
  std::mutex gate;
  std::thread::id owner;
  int depth;
  hipStream_t last_stream;   // compared only, never passed to a HIP call after its own call returned
  hipEvent_t done_event;     // recorded behind the work of every outermost call once two streams have been seen
  bool done_recorded;
  bool had_work, multi_stream;
};


namespace qlamd {
namespace rt {

// Per-robot run-time indexed arrays in LDS, [element][robot-in-wave]: a lane's
// bank depends on the lane only, so divergent element indices never conflict.
struct LdsScratch {
  double *base;
  int stride;
  __device__ __forceinline__ double &at(int e) { return base[e * stride]; }
};

struct LdsTab { // one leg's 64-double block of the model table, staged in LDS
  const double *p;
  __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};


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


// Entry guard of every call that uses the context.  A second thread entering while a call is in progress gets
// QLAMD_ERR_BUSY (the whole tick nests calls on its own thread: allowed).  Calls on one stream are ordered by the
// stream and cost nothing here.  When the stream changes, the new call's work must not overtake the previous call's
// (both use the context's scratch memory).  No handle of the caller's is kept beyond its call -- only its value, to see
// that the stream changed.  The first time that happens the device is drained once (the earlier stream cannot be named
// any more) and the context turns to event ordering: from then on every outermost call records the context's own event
// behind its work and a call on another stream makes its stream wait for it -- asynchronous, on the device.  A context
// that stays on one stream never records anything (an event record costs an eager launch loop ~3 us per call,
// measured).  While a stream is being captured into a graph neither is done (an event recorded inside a capture
// cannot be waited for outside it, and the reverse): the capturing caller orders the graph.
struct CallGuard {
  qlamd_context *c;
  int rc;
  hipStream_t st;
  bool uses_stream;
  static bool capturing(hipStream_t s) {
    // (the query itself counts as an unsafe call when `s` is the legacy stream and some other stream is being captured in
    // global mode -- it would invalidate that capture; in relaxed mode it is a plain query)
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    hipStreamCaptureStatus a = hipStreamCaptureStatusNone;
    const bool ok = hipStreamIsCapturing(s, &a) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    return ok && a != hipStreamCaptureStatusNone;
  }
  CallGuard(qlamd_context *ctx, hipStream_t stream, bool uses = true) : c(ctx), rc(QLAMD_OK), st(stream), uses_stream(uses) {
    const std::thread::id me = std::this_thread::get_id();
    {
      std::lock_guard<std::mutex> lk(c->gate);
      if (c->depth > 0 && c->owner != me) { rc = QLAMD_ERR_BUSY; c = nullptr; return; }
      c->owner = me;
      if (c->depth++ > 0) { uses_stream = false; return; } // nested: the outermost call does the ordering
    }
    if (!uses_stream || !c->had_work || c->last_stream == st || capturing(st)) return;
    if (!c->multi_stream) {
      // first change of stream: drain the device once (legal whatever another thread is capturing), then order by events
      c->multi_stream = true;
      hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
      (void)hipThreadExchangeStreamCaptureMode(&mode);
      if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
      (void)hipThreadExchangeStreamCaptureMode(&mode);
    } else if (c->done_recorded) {
      if (hipStreamWaitEvent(st, c->done_event, 0) != hipSuccess) (void)hipGetLastError();
    }
  }
  ~CallGuard() {
    if (!c) return;
    if (uses_stream) {
      // A call whose stream is being captured queues nothing now: it is invisible to the eager ordering state (the event
      // of the last eager call stays the one to wait for, and its stream the one compared with).
      // (asked whenever the answer matters: not for a call on the stream of the last eager call of a one-stream context,
      // whose state this leaves as it is either way)
      bool captured = false;
      if (c->multi_stream || !c->had_work || st != c->last_stream) captured = capturing(st);
      if (c->multi_stream) {
        if (!captured) {
          // mark the end of this call's work for a later call on another stream
          if (!c->done_event && hipEventCreateWithFlags(&c->done_event, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            c->done_event = nullptr;
          }
          c->done_recorded = c->done_event && hipEventRecord(c->done_event, st) == hipSuccess;
          if (!c->done_recorded) (void)hipGetLastError();
        }
      }
      if (!captured) {
        c->last_stream = st;
        c->had_work = true;
      }
    }
    std::lock_guard<std::mutex> lk(c->gate);
    c->depth--;
  }
};
#define QL_ENTER(ctx, st)                         \
  ::qlamd::rt::CallGuard ql_guard_((ctx), (st));  \
  if (ql_guard_.rc != QLAMD_OK) return ql_guard_.rc
// a call that queues nothing on a stream (options)
#define QL_ENTER_NO_STREAM(ctx)                            \
  ::qlamd::rt::CallGuard ql_guard_((ctx), nullptr, false); \
  if (ql_guard_.rc != QLAMD_OK) return ql_guard_.rc

// balance_kernel.hip: the control step behind qlamd_balance_solve_batch / qlamd_force_distribution_batch, with the
// whole tick's per-robot `live` flags (device pointer or NULL)
// Robots per wavefront of the balance kernel.  The lane-cooperative kernel (4) wins at every batch size measured
// (1 K ... 1 M robots, static and trot: tools/batch_sweep.py); the one-lane-per-robot kernels stay selectable as an
// independent second implementation (different QP linear algebra) for cross-checks.
inline int pick_rpw(const qlamd_context *ctx, int64_t batch) {
  (void)batch;
  return ctx->rpw_override ? ctx->rpw_override : 4;
}
int balance_impl(qlamd_context *ctx, const qlamd_state_batch *in_user, const double *wrench, const uint8_t *live, int support_only,
                 int64_t batch, double *joint_effort, double *contact_force, int32_t *status, int memory, void *stream,
                 const qlamd_placement *pl = nullptr);

// qlamd_placement_from_iterations on device pointers, for entries that hold the context's guard already (balance_kernel.hip)
int placement_launch(qlamd_context *ctx, const int32_t *d_iterations, int64_t batch, int policy, int32_t *d_order, hipStream_t st);

// The placement a lane-cooperative QP kernel runs in (qlamd_place_next_call): which problem sits in which slot of the
// launch, and where the iteration counts go.  Both NULL = slot s takes problem s.
// words of the context's place_sync block behind the shadow barrier's two: event counters (qlamd_get_counter)
constexpr int kSyncGiveUps = 2, kSyncWarmRetries = 3;
struct PlacePtrs {
  const int32_t *order;
  int32_t *iterations;
  // warm start (the whole-body step only: 64 bits per problem, i.e. [B][2] words of qlamd_placement's uint32 arrays)
  const unsigned long long *prev_working_set;
  unsigned long long *working_set;
  uint32_t *warm_retries; // the context's count of rejected warm starts, or NULL
  int32_t *identity_out = nullptr; // QLAMD_PLACEMENT_NONE with a warm start: every slot writes its own index here (the next call's order)
};
#ifdef __HIPCC__
// problem index of slot `slot` (row slot % 4 of wavefront slot / 4); live = the slot holds a problem (an order entry
// outside [0, B) leaves it empty); dead slots compute on problem B - 1 and write nothing
__device__ __forceinline__ int64_t placed_index(const PlacePtrs &pp, int64_t slot, int64_t B, bool &live) {
  live = slot < B;
  int64_t i = live ? slot : B - 1;
  if (pp.identity_out && live && (threadIdx.x & 15) == 0) pp.identity_out[slot] = (int32_t)slot;
  if (pp.order) {
    const int64_t o = pp.order[i];
    live = live && o >= 0 && o < B;
    i = live ? o : B - 1;
  }
  return i;
}
#endif
// Takes the pending placement of qlamd_place_next_call, if any, for a QLAMD_MEM_DEVICE call of `batch` problems.
// Returns QLAMD_OK and fills pp / *next (next->next_robot_order != NULL when a following placement was asked for), or
// QLAMD_ERR_INVALID_ARGUMENT for a host-memory call (the placement's arrays are device arrays).
inline int take_placement(qlamd_context *ctx, int memory, int64_t batch, PlacePtrs *pp, qlamd_placement *next) {
  *pp = PlacePtrs{nullptr, nullptr, nullptr, nullptr, (uint32_t *)ctx->place_sync + kSyncWarmRetries};
  memset(next, 0, sizeof(*next));
  if (!ctx->has_next_placement) return QLAMD_OK;
  const qlamd_placement pl = ctx->next_placement;
  ctx->has_next_placement = false;
  if (memory != QLAMD_MEM_DEVICE || batch > INT32_MAX) return QLAMD_ERR_INVALID_ARGUMENT;
  pp->order = pl.robot_order;
  pp->iterations = pl.iterations;
  pp->prev_working_set = reinterpret_cast<const unsigned long long *>(pl.prev_working_set);
  pp->working_set = reinterpret_cast<unsigned long long *>(pl.working_set);
  *next = pl;
  // QLAMD_PLACEMENT_AUTO with a warm start (the whole-body step): no placement up to 4096 problems, the throughput policy above --
  // include/qlamd.h; with no placement the solving launch writes the identity itself and nothing is launched behind it (the
  // placement's two launches were 5 of the 23 us of a warm-started whole-body step of 4096 robots)
  if (pl.next_robot_order && pl.policy == QLAMD_PLACEMENT_AUTO && (pl.prev_working_set || pl.working_set))
    next->policy = batch <= 4096 ? QLAMD_PLACEMENT_NONE : QLAMD_PLACEMENT_THROUGHPUT;
  if (pl.next_robot_order && next->policy == QLAMD_PLACEMENT_NONE && (pl.prev_working_set || pl.working_set)) {
    pp->identity_out = pl.next_robot_order;
    next->next_robot_order = nullptr;
  }
  return QLAMD_OK;
}
inline int finish_placement(qlamd_context *ctx, const qlamd_placement &pl, int64_t batch, hipStream_t st) {
  if (!pl.next_robot_order) return QLAMD_OK;
  return placement_launch(ctx, pl.prev_iterations, batch, pl.policy, pl.next_robot_order, st);
}

inline int ensure_ws(qlamd_context *ctx, size_t bytes) {
  if (ctx->ws_bytes >= bytes) return QLAMD_OK;
  if (ctx->ws) (void)hipFree(ctx->ws);
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  if (hipMalloc(&ctx->ws, bytes) != hipSuccess) return QLAMD_ERR_OUT_OF_MEMORY;
  ctx->ws_bytes = bytes;
  return QLAMD_OK;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline int ensure_pinned(qlamd_context *ctx, size_t bytes) {
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


} // namespace rt
} // namespace qlamd
