// The head of a launch shaped like the balance kernel's (1024 workgroups of one wavefront, every lane reads one double
// through each of eleven per-field arrays and writes one): what the kernel-argument fetch costs in front of the first loads,
// and what the alternatives take off it.  Diagnostic tool, not product.
//   plain      eleven pointers in a by-value struct (what balance_coop_kernel takes): s_load of the arguments, then the loads
//   scalars    the same eleven pointers as separate kernel arguments (no preload asked for)
//   preload    separate arguments compiled with -mllvm -amdgpu-kernarg-preload-count=16: the first 14 user SGPRs (seven
//              pointers) are in registers when the wavefront starts, the other four still come by s_load
//   compact    one base pointer and ten 32-bit offsets (in doubles) from it: 12 SGPRs, all preloaded
//   packed     ONE array of 40-double records (320 B per robot): one pointer, 2.5 cache lines per robot instead of ten
//              partially used ones
// Each variant: microseconds per launch in a hipGraph of 200 launches (events), cold L2 as in the real thing (the launches
// write and read 1.3 MB + 0.4 MB).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o launch_head launch_head.hip                              (plain, scalars, packed)
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16 -DPRELOAD -o launch_head_preload launch_head.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

struct Ptrs { const double *p[11]; };
static constexpr int kW[11] = {12, 3, 4, 3, 3, 3, 4, 3, 3, 1, 1}; // doubles per robot and field (the last two stand for flags)

__device__ __forceinline__ int field_index(int f, int lane16, int robot) {
  const int w = f == 0 ? 12 : (f == 2 || f == 6) ? 4 : (f >= 9 ? 1 : 3);
  return robot * w + (lane16 < w ? lane16 : 0);
}

__global__ __launch_bounds__(64) void k_plain(const Ptrs s, int B, double *__restrict__ out) {
  const int robot = blockIdx.x * 4 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  double acc = 0.0;
#pragma unroll
  for (int f = 0; f < 11; f++) acc += s.p[f][field_index(f, l, robot)];
  if (l < 12) out[12 * robot + l] = acc;
}
__global__ __launch_bounds__(64) void k_scalars(const double *p0, const double *p1, const double *p2, const double *p3,
                                                const double *p4, const double *p5, const double *p6, const double *p7,
                                                const double *p8, const double *p9, const double *p10, int B,
                                                double *__restrict__ out) {
  const int robot = blockIdx.x * 4 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const double *p[11] = {p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10};
  double acc = 0.0;
#pragma unroll
  for (int f = 0; f < 11; f++) acc += p[f][field_index(f, l, robot)];
  if (l < 12) out[12 * robot + l] = acc;
}
__global__ __launch_bounds__(64) void k_compact(const double *base, uint32_t o1, uint32_t o2, uint32_t o3, uint32_t o4, uint32_t o5,
                                                uint32_t o6, uint32_t o7, uint32_t o8, uint32_t o9, uint32_t o10,
                                                double *__restrict__ out, int B) {
  const int robot = blockIdx.x * 4 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const uint32_t o[11] = {0, o1, o2, o3, o4, o5, o6, o7, o8, o9, o10};
  double acc = 0.0;
#pragma unroll
  for (int f = 0; f < 11; f++) acc += (base + o[f])[field_index(f, l, robot)];
  if (l < 12) out[12 * robot + l] = acc;
}
__global__ __launch_bounds__(64) void k_packed(const double *rec, int B, double *__restrict__ out) {
  const int robot = blockIdx.x * 4 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const double *r = rec + 40 * (size_t)robot;
  // 40 doubles over 16 lanes: three loads per lane cover it (the third clamped)
  const double acc = r[l] + r[16 + l] + r[l < 8 ? 32 + l : 39];
  if (l < 12) out[12 * robot + l] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class F>
static double timed(F launch, hipStream_t st) {
  hipGraph_t g; hipGraphExec_t ge;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeGlobal) != hipSuccess) return -1.0;
  for (int k = 0; k < 200; k++) launch();
  if (hipStreamEndCapture(st, &g) != hipSuccess || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) return -1.0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 1e9;
  for (int r = 0; r < 9; r++) {
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (ms * 5.0 < best) best = ms * 5.0; // us per launch
  }
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return best;
}

int main() {
  const int B = 4096;
  size_t total = 0, off[11];
  for (int f = 0; f < 11; f++) { off[f] = total; total += (size_t)B * kW[f] + 32; } // (+32: fields are separate allocations' worth apart)
  double *base = nullptr, *out = nullptr, *rec = nullptr;
  CK(hipMalloc((void **)&base, total * 8)); CK(hipMemset(base, 0, total * 8));
  CK(hipMalloc((void **)&out, (size_t)B * 12 * 8)); CK(hipMalloc((void **)&rec, (size_t)B * 40 * 8)); CK(hipMemset(rec, 0, (size_t)B * 40 * 8));
  hipStream_t st; CK(hipStreamCreate(&st));
  Ptrs s;
  for (int f = 0; f < 11; f++) s.p[f] = base + off[f];
  const dim3 grid(B / 4), block(64);
#ifdef PRELOAD
  std::printf("compiled with -amdgpu-kernarg-preload-count=16\n");
#endif
  std::printf("%-10s %.2f us per launch\n", "plain", timed([&] { hipLaunchKernelGGL(k_plain, grid, block, 0, st, s, B, out); }, st));
  std::printf("%-10s %.2f us per launch\n", "scalars", timed([&] {
    hipLaunchKernelGGL(k_scalars, grid, block, 0, st, s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], s.p[5], s.p[6], s.p[7], s.p[8], s.p[9], s.p[10], B, out); }, st));
  std::printf("%-10s %.2f us per launch\n", "compact", timed([&] {
    hipLaunchKernelGGL(k_compact, grid, block, 0, st, (const double *)base, (uint32_t)off[1], (uint32_t)off[2], (uint32_t)off[3], (uint32_t)off[4],
                       (uint32_t)off[5], (uint32_t)off[6], (uint32_t)off[7], (uint32_t)off[8], (uint32_t)off[9], (uint32_t)off[10], out, B); }, st));
  std::printf("%-10s %.2f us per launch\n", "packed", timed([&] { hipLaunchKernelGGL(k_packed, grid, block, 0, st, (const double *)rec, B, out); }, st));
  return 0;
}
