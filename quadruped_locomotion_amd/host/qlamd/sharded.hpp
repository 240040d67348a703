// The balance solve sharded over several GPUs, from C++: one process per GPU, one qlamd context per device, contiguous
// shards of the global batch, the joint efforts of all shards collected on every rank by ONE RCCL all-gather per control
// step (or per G steps) -- BASELINE.json's north star: "host code stays C++ ... shard the batch across 8xMI355X with RCCL
// all-gather over xGMI only for result collection".  It replaces the loop of the reference's real-time manager, which
// calls update() of ONE controller every 2.5 ms (balance_controller_manager.cpp:48-57), by one call per step that solves
// this rank's robots and collects everybody's.  Needs qlamd.h, the HIP runtime and rccl.h; no torch, no Python.
//
//   shard_of / slot_of      which robots a rank owns and how large an all-gather slot is
//   publish_id / await_id   rendezvous through a file: rank 0 creates the communicator id, the others wait for it
//   ShardedBalanceSolver    context + two effort buffers + two streams + events + communicator; step() queues one control
//                           step, the all-gather of step k overlapping the solve of step k + 1 on the other buffer
// tests/cpp/multi_gpu_demo.cpp is a thin caller; INTEGRATION.md section 3b shows the loop.
#pragma once

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include "qlamd.h"

namespace qlamd {
namespace host {

struct Shard { int64_t first, count; };

// Contiguous shards; the remainder of robots / ranks goes to the last rank.  (The all-gather needs equal counts: every
// rank's slot is sized for the largest shard, slot_of.)
inline Shard shard_of(int rank, int ranks, int64_t robots) {
  const int64_t per = robots / ranks;
  Shard s{per * rank, per};
  if (rank == ranks - 1) s.count = robots - s.first;
  return s;
}
inline int64_t slot_of(int ranks, int64_t robots) { return shard_of(ranks - 1, ranks, robots).count; }

// Rendezvous through a file both sides can see.  Rank 0 writes the blob to `path`.tmp and renames it (atomically: a reader
// sees nothing or all of it); the others poll for the file.  Returns 0, or -1 on an I/O error / timeout.
inline int publish_id(const std::string &path, const void *id, size_t bytes) {
  const std::string tmp = path + ".tmp";
  std::FILE *g = std::fopen(tmp.c_str(), "wb");
  if (!g) return -1;
  const bool ok = std::fwrite(id, bytes, 1, g) == 1;
  if (std::fclose(g) != 0 || !ok) return -1;
  return std::rename(tmp.c_str(), path.c_str()) == 0 ? 0 : -1;
}
inline int await_id(const std::string &path, void *id, size_t bytes, int timeout_ms = 60000) {
  std::FILE *g = nullptr;
  for (int waited = 0; !(g = std::fopen(path.c_str(), "rb")); waited += 50) {
    if (waited >= timeout_ms) return -1;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
  const bool ok = std::fread(id, bytes, 1, g) == 1;
  std::fclose(g);
  return ok ? 0 : -1;
}
// Both in one: the id every rank ends up with (rank 0's).  `make` fills the blob on rank 0 (ncclGetUniqueId for the real
// thing; tests pass a pattern).
template <class Make>
inline int exchange_id(int rank, const std::string &path, void *id, size_t bytes, Make make) {
  if (rank == 0) {
    if (make(id) != 0) return -1;
    return path.empty() ? 0 : publish_id(path, id, bytes);
  }
  return await_id(path, id, bytes);
}

// One rank of the sharded solve.  Not copyable; destroy() (or the destructor) releases everything.
class ShardedBalanceSolver {
 public:
  struct Options {
    int gather_every = 1;  // control steps per all-gather (G): the efforts of G steps travel with one collective
    bool placed = true;    // qlamd_balance_solve_placed_batch with the caller's loop of qlamd.h (placement of the next
                           // step made in the shadow of this one) instead of the plain entry
    bool warm = false;     // with `placed`: every robot's active-set loop starts from the working set it ended with on the
                           // step before (qlamd_placement::prev_working_set / working_set, one array updated in place)
  };
  ShardedBalanceSolver() = default;
  ShardedBalanceSolver(const ShardedBalanceSolver &) = delete;
  ShardedBalanceSolver &operator=(const ShardedBalanceSolver &) = delete;
  ~ShardedBalanceSolver() { destroy(); }

  // 0, or the failing layer's code in *what ("hip", "rccl", "qlamd") and a negative value.
  int init(int rank, int ranks, int device, int64_t robots, const qlamd_balance_params &params, const ncclUniqueId &id,
           const Options &opt) {
    rank_ = rank; ranks_ = ranks; opt_ = opt;
    shard_ = shard_of(rank, ranks, robots);
    slot_ = slot_of(ranks, robots);
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    const int rc = qlamd_context_create(&params, nullptr, device, &ctx_);
    if (rc != QLAMD_OK) { err_ = qlamd_strerror(rc); return rc == QLAMD_ERR_NO_DEVICE ? -2 : -1; }
    if (ncclCommInitRank(&comm_, ranks, id, rank) != ncclSuccess) return fail("ncclCommInitRank");
    have_comm_ = true;
    // (a shard may be empty -- fewer robots than ranks: its arrays still exist, one element long)
    const size_t shard_doubles = shard_doubles_(), B = (size_t)(shard_.count > 0 ? shard_.count : 1);
    // the streams first: everything this object ever queues on the context goes to s_solve_, so the context sees one stream and
    // never has to order two (a first call on the null stream would make every later call record an event: ~3 us per step)
    if (hipStreamCreateWithFlags(&s_solve_, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s_gather_, hipStreamNonBlocking) != hipSuccess)
      return fail("hipStreamCreate");
    for (int b = 0; b < 2; ++b) {
      // Result collection IN PLACE: a step writes its efforts straight into this rank's block of the gathered buffer, and the
      // all-gather is called with sendbuff = recvbuff + rank * count -- RCCL's in-place form, which moves nothing locally (out of
      // place, one rank's "gather" is a copy kernel plus two event edges per step: +12 us per step from C++, round 5)
      if (hipMalloc((void **)&all_[b], (shard_doubles > 0 ? shard_doubles : 1) * ranks * 8) != hipSuccess ||
          hipMemset(all_[b], 0, (shard_doubles > 0 ? shard_doubles : 1) * ranks * 8) != hipSuccess)
        return fail("hipMalloc");
      tau_[b] = all_[b] + (size_t)rank * shard_doubles;
      if (opt.placed) {
        if (hipMalloc((void **)&order_[b], B * 4) != hipSuccess || hipMalloc((void **)&iters_[b], B * 4) != hipSuccess ||
            hipMemset(iters_[b], 0, B * 4) != hipSuccess)
          return fail("hipMalloc");
        // identity order to start from: the placement of zero counts (every robot equally hard, ties by index)
        if (qlamd_placement_from_iterations(ctx_, iters_[b], shard_.count, QLAMD_PLACEMENT_THROUGHPUT, order_[b], QLAMD_MEM_DEVICE,
                                            s_solve_) != QLAMD_OK)
          return fail("qlamd_placement_from_iterations");
      }
      if (opt.placed && opt.warm && b == 0 &&
          (hipMalloc((void **)&working_set_, B * 4) != hipSuccess || hipMemset(working_set_, 0, B * 4) != hipSuccess))
        return fail("hipMalloc");
      if (hipEventCreateWithFlags(&solved_[b], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&gathered_[b], hipEventDisableTiming) != hipSuccess)
        return fail("hipEventCreate");
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail("hipDeviceSynchronize");
    if (hipMalloc((void **)&status_, B * 4) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void **)&barrier_word_, 4) != hipSuccess || hipMemset(barrier_word_, 0, 4) != hipSuccess) return fail("hipMalloc");
    step_ = 0;
    pending_[0] = pending_[1] = false;
    return 0;
  }

  // One control step of this rank's shard (`in`: device pointers to the shard's state arrays), queued on the solve stream;
  // with_gather: the efforts of the group of G steps this one completes are all-gathered on the gather stream, overlapping
  // the following solves, which write the other buffer.  Call finish_group() before a sequence ends on an incomplete group.
  int step(const qlamd_state_batch &in, bool with_gather) {
    const int G = opt_.gather_every, group = (int)(step_ / G), b = group & 1, within = (int)(step_ % G);
    if (within == 0 && pending_[b]) {  // the gather of two groups ago has read buffer b
      if (hipStreamWaitEvent(s_solve_, gathered_[b], 0) != hipSuccess) return fail("hipStreamWaitEvent");
      pending_[b] = false;
    }
    double *out = tau_[b] + (size_t)within * slot_ * 12;
    int rc;
    if (opt_.placed) {
      const int k = (int)(step_ & 1);
      const qlamd_placement pl{order_[k], iters_[k], iters_[k ^ 1], order_[k ^ 1], QLAMD_PLACEMENT_AUTO, working_set_, working_set_};
      rc = qlamd_balance_solve_placed_batch(ctx_, &in, shard_.count, &pl, out, nullptr, status_, QLAMD_MEM_DEVICE, s_solve_);
    } else {
      rc = qlamd_balance_solve_batch(ctx_, &in, shard_.count, out, nullptr, status_, QLAMD_MEM_DEVICE, s_solve_);
    }
    if (rc != QLAMD_OK) { err_ = qlamd_strerror(rc); return -1; }
    ++step_;
    if (with_gather && within == G - 1) return gather_(b);
    return 0;
  }
  // all-gather of a group that a sequence leaves incomplete (its last step was not the G-th of its group)
  int finish_group(bool with_gather) {
    const int G = opt_.gather_every;
    if (!with_gather || step_ % G == 0) return 0;
    const int b = (int)(step_ / G) & 1;
    step_ += G - step_ % G; // the next step starts a new group, on the other buffer
    return gather_(b);
  }
  int drain() {
    if (hipStreamSynchronize(s_solve_) != hipSuccess || hipStreamSynchronize(s_gather_) != hipSuccess) return fail("hipStreamSynchronize");
    return 0;
  }

  // K control steps as ONE hipGraph: the steps are queued exactly as step() queues them -- solves on one stream, the all-gathers
  // on the other, joined by events -- while the solve stream is being captured, so that a replay costs the host one call instead
  // of five per step (an eager step from C++ is host-bound: launch, event record, stream wait, all-gather call, event record:
  // 38 us per step for a 27 us solve of 8192 robots, profiles/r6/multi_gpu_cpp_one_rank.txt).  `in` returns the states of step k
  // (device pointers that stay valid and are re-read by every replay: a caller that steps a trajectory rewrites them between
  // replays, or hands K different batches).  K should be even (the placed loop uses its two buffers in turn) and a multiple of
  // gather_every.  RCCL's collectives are capturable; the context's own ordering state is left alone by captured calls
  // (include/qlamd.h, "Threads and streams").  replay_captured() queues the graph on the solve stream; drain() waits for it.
  template <class States>
  int capture_steps(int K, bool with_gather, States in) {
    if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
    reset_steps();
    if (hipStreamBeginCapture(s_solve_, hipStreamCaptureModeThreadLocal) != hipSuccess) return fail("hipStreamBeginCapture");
    int rc = 0;
    for (int k = 0; k < K && rc == 0; ++k) rc = step(in(k), with_gather);
    if (rc == 0) rc = finish_group(with_gather);
    // join: the solve stream ends behind every gather still in flight
    for (int b = 0; b < 2 && rc == 0; ++b)
      if (pending_[b]) {
        if (hipStreamWaitEvent(s_solve_, gathered_[b], 0) != hipSuccess) rc = fail("hipStreamWaitEvent");
        pending_[b] = false;
      }
    hipGraph_t graph = nullptr;
    if (hipStreamEndCapture(s_solve_, &graph) != hipSuccess || !graph) return rc != 0 ? rc : fail("hipStreamEndCapture");
    if (rc == 0 && hipGraphInstantiate(&graph_exec_, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail("hipGraphInstantiate");
    (void)hipGraphDestroy(graph);
    captured_steps_ = rc == 0 ? K : 0;
    return rc;
  }
  int replay_captured() {
    if (!graph_exec_) return fail("no captured steps");
    if (hipGraphLaunch(graph_exec_, s_solve_) != hipSuccess) return fail("hipGraphLaunch");
    return 0;
  }
  int captured_steps() const { return captured_steps_; }
  // a barrier over all ranks (a one-element all-reduce of a word of its own on the gather stream), then drained
  int barrier() {
    if (ncclAllReduce(barrier_word_, barrier_word_, 1, ncclInt32, ncclMax, comm_, s_gather_) != ncclSuccess) return fail("ncclAllReduce");
    return drain();
  }
  void reset_steps() { step_ = 0; pending_[0] = pending_[1] = false; }

  const Shard &shard() const { return shard_; }
  int64_t slot() const { return slot_; }
  size_t gathered_doubles() const { return shard_doubles_() * ranks_; }
  // device buffer [ranks][G][slot][12] of gather group `group` (valid after drain())
  const double *gathered(int64_t group) const { return all_[group & 1]; }
  const int32_t *status() const { return status_; }
  qlamd_context *context() const { return ctx_; }
  const char *error() const { return err_; }

  void destroy() {
    if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
    if (have_comm_) { ncclCommDestroy(comm_); have_comm_ = false; }
    for (int b = 0; b < 2; ++b) {
      if (all_[b]) (void)hipFree(all_[b]); // (tau_[b] points into it)
      if (order_[b]) (void)hipFree(order_[b]);
      if (iters_[b]) (void)hipFree(iters_[b]);
      if (solved_[b]) (void)hipEventDestroy(solved_[b]);
      if (gathered_[b]) (void)hipEventDestroy(gathered_[b]);
      tau_[b] = all_[b] = nullptr; order_[b] = iters_[b] = nullptr; solved_[b] = gathered_[b] = nullptr;
    }
    if (status_) { (void)hipFree(status_); status_ = nullptr; }
    if (barrier_word_) { (void)hipFree(barrier_word_); barrier_word_ = nullptr; }
    if (working_set_) { (void)hipFree(working_set_); working_set_ = nullptr; }
    if (s_solve_) { (void)hipStreamDestroy(s_solve_); s_solve_ = nullptr; }
    if (s_gather_) { (void)hipStreamDestroy(s_gather_); s_gather_ = nullptr; }
    if (ctx_) { qlamd_context_destroy(ctx_); ctx_ = nullptr; }
  }

 private:
  size_t shard_doubles_() const { return (size_t)opt_.gather_every * slot_ * 12; }
  int gather_(int b) {
    if (hipEventRecord(solved_[b], s_solve_) != hipSuccess || hipStreamWaitEvent(s_gather_, solved_[b], 0) != hipSuccess)
      return fail("hipEventRecord");
    // in place: tau_[b] == all_[b] + rank * count
    if (ncclAllGather(tau_[b], all_[b], shard_doubles_(), ncclDouble, comm_, s_gather_) != ncclSuccess) return fail("ncclAllGather");
    if (hipEventRecord(gathered_[b], s_gather_) != hipSuccess) return fail("hipEventRecord");
    pending_[b] = true;
    return 0;
  }
  int fail(const char *what) { err_ = what; return -1; }

  int rank_ = 0, ranks_ = 1;
  Options opt_;
  Shard shard_{0, 0};
  int64_t slot_ = 0, step_ = 0;
  qlamd_context *ctx_ = nullptr;
  ncclComm_t comm_{};
  bool have_comm_ = false;
  double *tau_[2] = {nullptr, nullptr}, *all_[2] = {nullptr, nullptr};
  int32_t *order_[2] = {nullptr, nullptr}, *iters_[2] = {nullptr, nullptr}, *status_ = nullptr, *barrier_word_ = nullptr;
  uint32_t *working_set_ = nullptr;
  hipStream_t s_solve_ = nullptr, s_gather_ = nullptr;
  hipEvent_t solved_[2] = {nullptr, nullptr}, gathered_[2] = {nullptr, nullptr};
  bool pending_[2] = {false, false};
  hipGraphExec_t graph_exec_ = nullptr;
  int captured_steps_ = 0;
  const char *err_ = "";
};

} // namespace host
} // namespace qlamd
