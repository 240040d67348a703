// The sharded balance solve driven from C++: one process per GPU, a qlamd context per device, contiguous shards of the
// global batch, and the joint efforts of all shards collected on every rank with ONE RCCL all-gather per control step --
// the host side BASELINE's north star asks for ("host code stays C++ ... shard the batch across 8xMI355X with RCCL
// all-gather over xGMI only for result collection"), without torch: qlamd.h, the HIP runtime and rccl.h only.
//
//   multi_gpu_demo --states FILE --robots N [--rank R --ranks W --id-file F] [--steps K] [--out FILE] [--gather-every G]
//   multi_gpu_demo --selftest-sharding          (no GPU: prints the shard of every rank for a few batch sizes)
//
// --states: the global batch as written by tests/test_multi_gpu_cpp.py -- ten float64 arrays [N][k] one after the other
//   in the order of qlamd_state_batch (k = 12 3 4 3 3 3 4 3 3) followed by support_leg [N][4] uint8.
// Rank r solves robots [r * B, (r + 1) * B) with B = N / W (the last rank takes the remainder: shard_of below) on device
// LOCAL_RANK (default r) and owns row block r of the gathered [N][12] array.
// Rendezvous: rank 0 writes the ncclUniqueId to --id-file (atomically, by rename), the others wait for the file.
// Pipeline (as bench.py's hipGraph form, here with plain streams and events): the solve of step k writes torque buffer
// k & 1 on the solve stream; the gather of step k reads it on the gather stream and overlaps the solve of step k + 1,
// which writes the other buffer; the solve of step k + 2 waits for gather k before it reuses the buffer.
// With --gather-every G the efforts of G consecutive steps are collected by one all-gather of G * B * 96 bytes per rank.
// Exit codes: 0 ok, 2 usage, 3 no device, 4 a library call failed.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "qlamd.h"

namespace {

struct Shard { int64_t first, count; };

// contiguous shards; the remainder of N / W goes to the last rank (the all-gather needs equal counts: every rank's slot
// is ceil-sized, see slot_of)
Shard shard_of(int rank, int ranks, int64_t robots) {
  const int64_t per = robots / ranks;
  Shard s{per * rank, per};
  if (rank == ranks - 1) s.count = robots - s.first;
  return s;
}
int64_t slot_of(int ranks, int64_t robots) { // robots per all-gather slot: the largest shard
  return shard_of(ranks - 1, ranks, robots).count;
}

#define HIP_OK(call)                                                                                      \
  do {                                                                                                    \
    const hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 4; }  \
  } while (0)
#define NCCL_OK(call)                                                                                     \
  do {                                                                                                    \
    const ncclResult_t e_ = (call);                                                                       \
    if (e_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(e_)); return 4; } \
  } while (0)
#define QL_OK(call)                                                                                       \
  do {                                                                                                    \
    const int e_ = (call);                                                                                \
    if (e_ != QLAMD_OK) { std::fprintf(stderr, "%s: %s\n", #call, qlamd_strerror(e_)); return e_ == QLAMD_ERR_NO_DEVICE ? 3 : 4; } \
  } while (0)

int selftest_sharding() {
  for (int64_t robots : {int64_t(4096), int64_t(65536), int64_t(65537), int64_t(10)}) {
    for (int ranks : {1, 2, 3, 8}) {
      std::printf("shards %lld %d slot %lld :", (long long)robots, ranks, (long long)slot_of(ranks, robots));
      for (int r = 0; r < ranks; ++r) {
        const Shard s = shard_of(r, ranks, robots);
        std::printf(" %lld+%lld", (long long)s.first, (long long)s.count);
      }
      std::printf("\n");
    }
  }
  return 0;
}

} // namespace

int main(int argc, char **argv) {
  std::string states_path, out_path, id_path;
  int64_t robots = 0;
  int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0;
  int ranks = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
  int steps = 20, gather_every = 1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    const auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--selftest-sharding") return selftest_sharding();
    else if (a == "--states") states_path = next();
    else if (a == "--out") out_path = next();
    else if (a == "--id-file") id_path = next();
    else if (a == "--robots") robots = std::atoll(next());
    else if (a == "--rank") rank = std::atoi(next());
    else if (a == "--ranks") ranks = std::atoi(next());
    else if (a == "--steps") steps = std::atoi(next());
    else if (a == "--gather-every") gather_every = std::atoi(next());
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  if (states_path.empty() || robots <= 0 || ranks < 1 || rank < 0 || rank >= ranks || steps < 1 || gather_every < 1 ||
      (ranks > 1 && id_path.empty())) {
    std::fprintf(stderr, "usage: multi_gpu_demo --states FILE --robots N [--rank R --ranks W --id-file F] [--steps K] "
                         "[--gather-every G] [--out FILE] | --selftest-sharding\n");
    return 2;
  }
  const int device = std::getenv("LOCAL_RANK") ? std::atoi(std::getenv("LOCAL_RANK")) : rank;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) { std::fprintf(stderr, "no device %d\n", device); return 3; }
  HIP_OK(hipSetDevice(device));

  // ---- this rank's shard of the global batch, host -> device
  const Shard sh = shard_of(rank, ranks, robots);
  const int64_t slot = slot_of(ranks, robots), B = sh.count;
  static const int kWidth[9] = {12, 3, 4, 3, 3, 3, 4, 3, 3};
  std::FILE *f = std::fopen(states_path.c_str(), "rb");
  if (!f) { std::fprintf(stderr, "cannot open %s\n", states_path.c_str()); return 2; }
  double *dfield[9];
  uint8_t *dsupport = nullptr;
  int64_t offset = 0;
  for (int k = 0; k < 9; ++k) {
    std::vector<double> h((size_t)B * kWidth[k]);
    std::fseek(f, (long)(offset + sh.first * kWidth[k] * 8), SEEK_SET);
    if (std::fread(h.data(), 8, h.size(), f) != h.size()) { std::fprintf(stderr, "short read\n"); return 2; }
    offset += robots * kWidth[k] * 8;
    HIP_OK(hipMalloc((void **)&dfield[k], h.size() * 8));
    HIP_OK(hipMemcpy(dfield[k], h.data(), h.size() * 8, hipMemcpyHostToDevice));
  }
  {
    std::vector<uint8_t> h((size_t)B * 4);
    std::fseek(f, (long)(offset + sh.first * 4), SEEK_SET);
    if (std::fread(h.data(), 1, h.size(), f) != h.size()) { std::fprintf(stderr, "short read\n"); return 2; }
    HIP_OK(hipMalloc((void **)&dsupport, h.size()));
    HIP_OK(hipMemcpy(dsupport, h.data(), h.size(), hipMemcpyHostToDevice));
  }
  std::fclose(f);
  qlamd_state_batch in;
  std::memset(&in, 0, sizeof(in));
  in.joint_position = dfield[0]; in.base_position = dfield[1]; in.base_orientation = dfield[2];
  in.base_linear_velocity = dfield[3]; in.base_angular_velocity = dfield[4]; in.desired_position = dfield[5];
  in.desired_orientation = dfield[6]; in.desired_linear_velocity = dfield[7]; in.desired_angular_velocity = dfield[8];
  in.support_leg = dsupport;

  // ---- one context per device; RCCL communicator
  qlamd_balance_params params;
  qlamd_balance_default_params(&params);
  qlamd_context *ctx = nullptr;
  QL_OK(qlamd_context_create(&params, nullptr, device, &ctx));
  ncclUniqueId id;
  if (rank == 0) {
    NCCL_OK(ncclGetUniqueId(&id));
    if (!id_path.empty()) {
      const std::string tmp = id_path + ".tmp";
      std::FILE *g = std::fopen(tmp.c_str(), "wb");
      if (!g || std::fwrite(&id, sizeof(id), 1, g) != 1) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 4; }
      std::fclose(g);
      std::rename(tmp.c_str(), id_path.c_str());
    }
  } else {
    std::FILE *g = nullptr;
    for (int tries = 0; tries < 600 && !(g = std::fopen(id_path.c_str(), "rb")); ++tries)
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    if (!g || std::fread(&id, sizeof(id), 1, g) != 1) { std::fprintf(stderr, "no id in %s\n", id_path.c_str()); return 4; }
    std::fclose(g);
  }
  ncclComm_t comm;
  NCCL_OK(ncclCommInitRank(&comm, ranks, id, rank));

  // ---- buffers: two sets, each holding the efforts of `gather_every` steps of this shard (slot-sized rows, so that every
  // rank contributes the same count) and the gathered result [ranks][gather_every][slot][12]
  const size_t shard_doubles = (size_t)gather_every * slot * 12, all_doubles = shard_doubles * ranks;
  double *tau[2], *all[2];
  int32_t *status = nullptr;
  for (int b = 0; b < 2; ++b) {
    HIP_OK(hipMalloc((void **)&tau[b], shard_doubles * 8));
    HIP_OK(hipMemset(tau[b], 0, shard_doubles * 8));
    HIP_OK(hipMalloc((void **)&all[b], all_doubles * 8));
  }
  HIP_OK(hipMalloc((void **)&status, (size_t)B * 4));
  hipStream_t s_solve, s_gather;
  HIP_OK(hipStreamCreateWithFlags(&s_solve, hipStreamNonBlocking));
  HIP_OK(hipStreamCreateWithFlags(&s_gather, hipStreamNonBlocking));
  hipEvent_t solved[2], gathered[2], t0, t1;
  for (int b = 0; b < 2; ++b) {
    HIP_OK(hipEventCreateWithFlags(&solved[b], hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&gathered[b], hipEventDisableTiming));
  }
  HIP_OK(hipEventCreate(&t0));
  HIP_OK(hipEventCreate(&t1));

  const auto run = [&](int nsteps, bool with_gather) -> int {
    bool pending[2] = {false, false};
    for (int k = 0; k < nsteps; ++k) {
      const int group = k / gather_every, b = group & 1, within = k % gather_every;
      if (within == 0 && pending[b]) { HIP_OK(hipStreamWaitEvent(s_solve, gathered[b], 0)); pending[b] = false; }
      QL_OK(qlamd_balance_solve_batch(ctx, &in, B, tau[b] + (size_t)within * slot * 12, nullptr, status, QLAMD_MEM_DEVICE, s_solve));
      if (with_gather && (within == gather_every - 1 || k == nsteps - 1)) {
        HIP_OK(hipEventRecord(solved[b], s_solve));
        HIP_OK(hipStreamWaitEvent(s_gather, solved[b], 0));
        NCCL_OK(ncclAllGather(tau[b], all[b], shard_doubles, ncclDouble, comm, s_gather));
        HIP_OK(hipEventRecord(gathered[b], s_gather));
        pending[b] = true;
      }
    }
    HIP_OK(hipStreamSynchronize(s_solve));
    HIP_OK(hipStreamSynchronize(s_gather));
    return 0;
  };
  if (int rc = run(4, true)) return rc; // warm-up (RCCL sets its channels up on the first collective)

  double ms[2] = {0.0, 0.0};
  for (int with_gather = 1; with_gather >= 0; --with_gather) {
    // every rank starts together: a tiny all-reduce as barrier
    NCCL_OK(ncclAllReduce(status, status, 1, ncclInt32, ncclMax, comm, s_gather));
    HIP_OK(hipStreamSynchronize(s_gather));
    const auto w0 = std::chrono::steady_clock::now();
    if (int rc = run(steps, with_gather != 0)) return rc;
    ms[with_gather] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  }
  // the last gathered group, back on the host: [ranks][slot][12] of its last step
  const int last_group = (steps - 1) / gather_every, lb = last_group & 1, last_within = (steps - 1) % gather_every;
  if (!out_path.empty() && rank == 0) {
    // re-run with the gather so that all[lb] holds the last step (the timed no-gather loop ran last)
    if (int rc = run(steps, true)) return rc;
    std::vector<double> h(all_doubles);
    HIP_OK(hipMemcpy(h.data(), all[lb], all_doubles * 8, hipMemcpyDeviceToHost));
    std::FILE *g = std::fopen(out_path.c_str(), "wb");
    if (!g) { std::fprintf(stderr, "cannot write %s\n", out_path.c_str()); return 4; }
    for (int r = 0; r < ranks; ++r) {
      const Shard s = shard_of(r, ranks, robots);
      const double *src = h.data() + (size_t)r * shard_doubles + (size_t)last_within * slot * 12;
      std::fwrite(src, 8, (size_t)s.count * 12, g);
    }
    std::fclose(g);
  } else if (!out_path.empty()) {
    if (int rc = run(steps, true)) return rc; // the collective needs every rank
  }
  std::vector<int32_t> hs((size_t)B);
  HIP_OK(hipMemcpy(hs.data(), status, (size_t)B * 4, hipMemcpyDeviceToHost));
  int64_t failed = 0;
  for (int32_t v : hs) failed += v != QLAMD_STATUS_OK;
  std::printf("rank %d of %d device %d robots %lld+%lld steps %d gather_every %d : %.1f us/step with the all-gather, %.1f without, "
              "%lld robots with status != ok\n",
              rank, ranks, device, (long long)sh.first, (long long)B, steps, gather_every, 1e3 * ms[1] / steps, 1e3 * ms[0] / steps,
              (long long)failed);

  ncclCommDestroy(comm);
  qlamd_context_destroy(ctx);
  return 0;
}
