// Does the issue rate of a lone wavefront depend on how many of its four 16-lane rows are enabled in EXEC?
// (Found with tools/experiments/row_mix_probe.py: a wavefront of the balance kernel whose busy robots fill one or two rows
// runs every pass ~7 % slower than one with three or four.)  Same instruction patterns as issue_model.hip, measured with
// s_memtime inside a region whose execution mask is the given set of rows.  Diagnostic tool, not product.
//   hipcc --offload-arch=gfx950 -O2 -o exec_mask_model exec_mask_model.hip && ./exec_mask_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP16(REP4(x))
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

#define KERNEL(name, body)                                                                      \
  __global__ void name(unsigned long long *out, double *sink, int iters, unsigned rows) {       \
    double a = sink[threadIdx.x], b = sink[64 + threadIdx.x], c = sink[128 + threadIdx.x];      \
    double d = sink[192 + threadIdx.x], e = a + 1.0, f = b + 2.0;                               \
    float fa = (float)a, fb = (float)b;                                                         \
    int ia = (int)threadIdx.x, ib = ia * 3;                                                     \
    unsigned long long t0 = 0, t1 = 0;                                                          \
    if ((rows >> (threadIdx.x >> 4)) & 1u) {                                                    \
      STAMP(t0);                                                                                \
      for (int it = 0; it < iters; it++) { body }                                               \
      STAMP(t1);                                                                                \
    }                                                                                           \
    sink[256 + threadIdx.x] = a + b + c + d + e + f + fa + fb + ia + ib;                        \
    if ((threadIdx.x & 15) == 0 && ((rows >> (threadIdx.x >> 4)) & 1u)) out[0] = t1 - t0;       \
  }

KERNEL(k_fma_dep, REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));))
KERNEL(k_fma_ind4, REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                                      : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));))
KERNEL(k_add32_dep, REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));))
KERNEL(k_add32_ind, REP16(asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %2"
                                       : "+v"(fa), "+v"(fb), "+v"(ia), "+v"(ib) : "v"(fb));))
KERNEL(k_fmacdpp_dep, REP64(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));))
KERNEL(k_fmacdpp_ind3, REP16(asm volatile("v_fmac_f64_dpp %0, %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                                          "v_fmac_f64_dpp %1, %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                                          "v_fmac_f64_dpp %2, %3, %4 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                                          "v_fmac_f64_dpp %0, %3, %4 row_newbcast:9 row_mask:0xf bank_mask:0xf"
                                          : "+v"(a), "+v"(d), "+v"(e) : "v"(b), "v"(c));))
KERNEL(k_mov_dpp, REP64(asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(ia));))
KERNEL(k_cndmask, REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ib) : "vcc");))
KERNEL(k_nop0, REP64(asm volatile("s_nop 0");))
KERNEL(k_salu, REP64(asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");))
KERNEL(k_rcp, REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));))
KERNEL(k_bperm, REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(ia) : "v"(ib));))
KERNEL(k_mix, REP16(asm volatile("v_fma_f64 %0, %0, %3, %4\n\tv_cndmask_b32 %1, %1, %2, vcc\n\ts_add_u32 s20, s20, 1\n\tv_add_f32 %5, %5, %5"
                                 : "+v"(a), "+v"(ia), "+v"(ib) : "v"(b), "v"(c), "v"(fa) : "vcc", "s20", "scc");))

struct Entry { const char *name; void (*fn)(unsigned long long *, double *, int, unsigned); };

int main() {
  const int iters = 200;
  unsigned long long *out;
  double *sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, 4096 * sizeof(double));
  std::vector<double> h(4096, 1.0000001);
  hipMemcpy(sink, h.data(), 4096 * sizeof(double), hipMemcpyHostToDevice);
  Entry tab[] = {{"fma_f64 dependent", k_fma_dep}, {"fma_f64 4 independent", k_fma_ind4}, {"add_f32 dependent", k_add32_dep},
                 {"add_f32/u32 independent", k_add32_ind}, {"fmac_f64_dpp dependent", k_fmacdpp_dep},
                 {"fmac_f64_dpp 3 accumulators", k_fmacdpp_ind3}, {"s_nop1 + mov_b32_dpp", k_mov_dpp},
                 {"v_cndmask_b32 dependent", k_cndmask}, {"s_nop 0", k_nop0}, {"s_add_u32 dependent", k_salu},
                 {"v_rcp_f64 dependent", k_rcp}, {"ds_bpermute round trip", k_bperm}, {"fma64 / cndmask / salu / add32 mix", k_mix}};
  const unsigned masks[] = {0xF, 0x7, 0x3, 0x1, 0x8, 0x5};
  printf("%-38s", "cycles per instruction, rows enabled:");
  for (unsigned m : masks) printf("   0x%X  ", m);
  printf("\n");
  for (auto &e : tab) {
    printf("%-38s", e.name);
    for (unsigned m : masks) {
      for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(e.fn, dim3(1), dim3(64), 0, 0, out, sink, iters, m);
      hipDeviceSynchronize();
      unsigned long long r;
      hipMemcpy(&r, out, sizeof(r), hipMemcpyDeviceToHost);
      printf(" %7.2f ", (double)r / iters / 64.0);
    }
    printf("\n");
  }
  return 0;
}
