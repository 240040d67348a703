// Single-wave issue / latency model of gfx950 for the instruction mix of the balance kernel.
// One block of 64 lanes (one wavefront alone on its SIMD) runs N copies of an instruction pattern
// between two s_memtime stamps; prints shader-clock cycles per instruction.  Diagnostic tool, not product.
//   hipcc --offload-arch=gfx950 -O2 -o issue_model issue_model.hip && ./issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP16(REP4(x))

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

#define KERNEL(name, per_iter, body)                                                            \
  __global__ void name(unsigned long long *out, double *sink, int iters) {                      \
    double a = sink[threadIdx.x], b = sink[64 + threadIdx.x], c = sink[128 + threadIdx.x];      \
    double d = sink[192 + threadIdx.x], e = a + 1.0, f = b + 2.0, g = c + 3.0, h = d + 4.0;    \
    float fa = (float)a, fb = (float)b;                                                         \
    int ia = (int)threadIdx.x, ib = ia * 3;                                                     \
    unsigned long long m0 = 0x5555555555555555ull, m1 = 0x3333333333333333ull;                 \
    unsigned long long t0, t1;                                                                  \
    STAMP(t0);                                                                                  \
    for (int it = 0; it < iters; it++) { body }                                                 \
    STAMP(t1);                                                                                  \
    sink[256 + threadIdx.x] = a + b + c + d + e + f + g + h + fa + fb + ia + ib + (double)(m0 ^ m1); \
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = per_iter; } \
  }

// a: dependent v_fma_f64 chain
KERNEL(k_fma_dep, 64, REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));))
// b: four independent chains
KERNEL(k_fma_ind4, 64, REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                                           : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));))
// c: dependent f32 add
KERNEL(k_add32_dep, 64, REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));))
// d: the f64 row-reduction idiom: 2 mov_dpp + s_nop + add (4 instrs counted as 3 + nop)
KERNEL(k_red64, 64, REP16(asm volatile("v_mov_b32_dpp %1, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                                        "v_mov_b32_dpp %2, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                                        "v_add_f64 %4, %4, %5\n\ts_nop 1"
                                        : "+v"(ia), "+v"(ib), "+v"(ia), "+v"(ib), "+v"(a) : "v"(b));))
// d2: a true dependent reduction level: mov lo, mov hi (of a), add a += moved; the compiler-visible form
__device__ __forceinline__ double dpp_ror4(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x124, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x124, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
KERNEL(k_red64_dep, 64, REP16(a += dpp_ror4(a); asm volatile("" : "+v"(a));))  // 16 levels = 64 instrs incl. nops (approx.)
// e: v_fmac_f64_dpp dependent on one accumulator / three accumulators
KERNEL(k_fmacdpp_dep, 64, REP64(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));))
KERNEL(k_fmacdpp_ind3, 63, REP16(asm volatile("v_fmac_f64_dpp %0, %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                                               "v_fmac_f64_dpp %1, %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                                               "v_fmac_f64_dpp %2, %3, %4 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                                               : "+v"(a), "+v"(d), "+v"(e) : "v"(b), "v"(c));)
        REP4(REP4(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(f) : "v"(b), "v"(c));)))
// f: SALU chain
KERNEL(k_salu_dep, 64, REP64(asm volatile("s_and_b64 %0, %0, %1" : "+s"(m0) : "s"(m1));))
// g: alternating SALU / VALU (independent)
KERNEL(k_salu_valu, 64, REP16(REP4(asm volatile("s_xor_b64 %0, %0, %2\n\tv_fma_f64 %1, %1, %3, %4" : "+s"(m0), "+v"(a) : "s"(m1), "v"(b), "v"(c));)) )
// h: v_cndmask pair (f64 select) dependent
KERNEL(k_cndmask, 64, REP16(REP4(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ib) : "vcc");)))
// i: s_nop 1
KERNEL(k_nop1, 64, REP64(asm volatile("s_nop 1");))
KERNEL(k_nop0, 64, REP64(asm volatile("s_nop 0");))
// j: rcp seed
KERNEL(k_rcp, 64, REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));))
// k: cmp -> sgpr mask -> cndmask chain (3 instrs per link)
KERNEL(k_cmp_sel, 63, REP16(REP4(asm volatile("v_cmp_gt_f64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc\n\t" "v_add_f64 %0, %0, %1"
                                              : "+v"(a), "+v"(b), "+v"(ia) : "v"(ib) : "vcc");)))
// l: f32 dpp add reduction level (with the nop the compiler inserts)
KERNEL(k_add32_dpp, 64, REP64(asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(fa));))
// m: mov_dpp 32-bit dependent
KERNEL(k_mov_dpp, 64, REP64(asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(ia));))
// n: v_mov_b64 dpp newbcast
KERNEL(k_mov64_bc, 64, REP64(asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a));))
// o: ds_bpermute round trip
KERNEL(k_bperm, 64, REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(ia) : "v"(ib));))
// p: v_permlane32_swap / v_permlane16_swap (gfx950)
KERNEL(k_pl16swap, 64, REP64(asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(ia), "+v"(ib));))
KERNEL(k_pl32swap, 64, REP64(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ia), "+v"(ib));))
// q: v_readlane + s-> v
KERNEL(k_readlane, 64, REP16(REP4(asm volatile("v_readlane_b32 s20, %0, 5\n\tv_mov_b32 %0, s20" : "+v"(ia) : : "s20");)))
// r: dependent mul/add f64 mixed with independent f32 (co-issue?)
KERNEL(k_fma_f32_mix, 64, REP16(REP4(asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_add_f32 %1, %1, %1" : "+v"(a), "+v"(fa) : "v"(b), "v"(c));)))
// s: v_pk_fma_f32 dependent (packed two floats)
KERNEL(k_pkfma, 64, REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));))
// t: exec save/restore pair around a tiny block
KERNEL(k_saveexec, 64, REP16(REP4(asm volatile("s_and_saveexec_b64 s[20:21], %0\n\ts_or_b64 exec, exec, s[20:21]" : : "s"(m1) : "s20", "s21");)))
// u: ds_write + ds_read round trip (same lane)
KERNEL(k_lds_rt, 64, REP16(REP4(asm volatile("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(ib));)))

struct Entry { const char *name; void (*fn)(unsigned long long *, double *, int); const char *what; };

int main(int argc, char **argv) {
  int iters = 200;
  unsigned long long *out;
  double *sink;
  hipMalloc(&out, 2 * 4096 * sizeof(unsigned long long));
  hipMalloc(&sink, 4096 * sizeof(double));
  std::vector<double> h(4096, 1.0000001);
  hipMemcpy(sink, h.data(), 4096 * sizeof(double), hipMemcpyHostToDevice);
  Entry tab[] = {
      {"fma_f64 dependent", k_fma_dep, "v_fma_f64 chain"},
      {"fma_f64 4 independent", k_fma_ind4, ""},
      {"add_f32 dependent", k_add32_dep, ""},
      {"2 mov_dpp + add_f64 + s_nop1 (independent)", k_red64, "4 instrs per group, 16 groups"},
      {"a += ror4(a) compiled", k_red64_dep, "per level (16 levels per iter)"},
      {"fmac_f64_dpp dependent", k_fmacdpp_dep, ""},
      {"fmac_f64_dpp 3 accumulators", k_fmacdpp_ind3, ""},
      {"s_and_b64 dependent", k_salu_dep, ""},
      {"salu+valu alternating", k_salu_valu, "per pair"},
      {"v_cndmask_b32 dependent", k_cndmask, ""},
      {"s_nop 1", k_nop1, ""},
      {"s_nop 0", k_nop0, ""},
      {"v_rcp_f64 dependent", k_rcp, ""},
      {"cmp->cndmask->add chain", k_cmp_sel, "per 3 instrs"},
      {"s_nop1 + add_f32_dpp", k_add32_dpp, "per pair"},
      {"s_nop1 + mov_b32_dpp", k_mov_dpp, "per pair"},
      {"s_nop1 + mov_b64_dpp newbcast", k_mov64_bc, "per pair"},
      {"ds_bpermute round trip", k_bperm, ""},
      {"permlane16_swap", k_pl16swap, ""},
      {"permlane32_swap", k_pl32swap, ""},
      {"readlane + mov", k_readlane, "per pair"},
      {"fma_f64 + add_f32 pair", k_fma_f32_mix, "per pair"},
      {"pk_fma_f32 dependent", k_pkfma, ""},
      {"saveexec + restore", k_saveexec, "per pair"},
      {"lds write+read round trip", k_lds_rt, ""},
  };
  for (int blocks : {1, 1024}) {
    printf("== %d block(s) of 64 lanes, %d iterations\n", blocks, iters);
    for (auto &e : tab) {
      for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, out, sink, iters);
      hipDeviceSynchronize();
      unsigned long long r[2];
      hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
      double per_iter = (double)r[0] / iters;
      printf("%-46s %9.1f cycles/iter  %6.2f per unit (%s)\n", e.name, per_iter, per_iter / 64.0, e.what);
    }
  }
  // clock calibration: s_memtime vs wall clock over a long kernel
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_fma_dep, dim3(1), dim3(64), 0, 0, out, sink, 20000);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long r[2];
  hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
  printf("calibration: %llu s_memtime ticks in %.3f ms => %.1f MHz\n", r[0], ms, r[0] / (ms * 1e3));
  return 0;
}
