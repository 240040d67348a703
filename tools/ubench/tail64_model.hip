// The part of an active-set pass that a 64-lane form of a lone robot would change, in both forms, timed on one wavefront
// (DESIGN.md section 4.1).  What a pass does with H and N* -- directions z = H n_p, r = N* n_p, then the
// rank-one updates H += hc vec', N* += nc vec' with vectors made from z, r and 1 / z'n_p -- with the product's own
// primitives (csrc/coop_lanes.hpp) and the product's dependency chain (update -> next directions through H and N*):
//   form A (the library): a row of H and of N* per lane, all twelve columns; 24 + 24 broadcast-FMAs, three partial sums each.
//   form B (one robot on all four rows): row rho keeps the three columns of leg rho.  n_p and vec rotated by 4 rho lanes
//          (ds_bpermute, the broadcast lane of v_fmac_f64_dpp is the same for every row), 6 + 6 broadcast-FMAs, and an
//          all-reduce of the partial z and r over the four rows (v_permlane32_swap / v_permlane16_swap + adds).
// The rest of a pass (step lengths, selection: ~180 instructions) is the same in both forms and is left out, so the difference
// of the two figures is what form B would gain per pass, before the cost of re-laying a robot over the wavefront.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../quadruped_locomotion_amd/csrc -I../../include -o tail64_model tail64_model.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>
#include <vector>

#include "coop_lanes.hpp"

using namespace qlamd::coop;

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

__device__ __forceinline__ double rotated(double v, int addr) {
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// sum over the four rows, per lane position, result in every row
__device__ __forceinline__ double rows_sum(double v) {
  int a_lo = __double2loint(v), a_hi = __double2hiint(v), b_lo = a_lo, b_hi = a_hi;
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a_lo), "+v"(b_lo));
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a_hi), "+v"(b_hi));
  const double s = __hiloint2double(a_hi, a_lo) + __hiloint2double(b_hi, b_lo);
  a_lo = __double2loint(s); a_hi = __double2hiint(s); b_lo = a_lo; b_hi = a_hi;
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a_lo), "+v"(b_lo));
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a_hi), "+v"(b_hi));
  return __hiloint2double(a_hi, a_lo) + __hiloint2double(b_hi, b_lo);
}

__global__ void form_a(unsigned long long *out, double *io, int passes) {
  const int lane = threadIdx.x;
  double H[12], Ns[12];
#pragma unroll
  for (int j = 0; j < 12; j++) { H[j] = io[64 * j + lane]; Ns[j] = io[64 * (12 + j) + lane]; }
  double npj = io[64 * 24 + lane];
  const double zb = 0.0625;
  unsigned long long t0, t1;
  STAMP(t0);
  for (int p = 0; p < passes; p++) {
    double za[3] = {0.0, 0.0, 0.0}, ra[3] = {0.0, 0.0, 0.0};
    static_for<12>([&](auto J) {
      constexpr int j = J;
      fmac_bc<lane_of(j), j == 0>(za[j % 3], npj, H[j]);
      fmac_bc<lane_of(j)>(ra[j % 3], npj, Ns[j]);
    });
    const double z = (za[0] + za[1]) + za[2], r = (ra[0] + ra[1]) + ra[2];
    const double zn = row_sum(fma(z, npj, zb));
    const double zinv = rcp_nr(zn);
    const double vec = z * zinv, hc = -z, nc = -r;
    static_for<12>([&](auto J) {
      constexpr int j = J;
      fmac_bc<lane_of(j), j == 0>(H[j], vec, hc);
      fmac_bc<lane_of(j)>(Ns[j], vec, nc);
    });
  }
  STAMP(t1);
  double acc = npj;
#pragma unroll
  for (int j = 0; j < 12; j++) acc += H[j] + Ns[j];
  io[64 * 25 + lane] = acc;
  if (lane == 0) out[0] = t1 - t0;
}

// kHideNp: the rotation of n_p is left out of the pass (the best case: n_p is known at the end of the selection, whose own
// cross-lane waits could hide it; the rotation of vec sits on the critical path whatever is done)
template <bool kHideNp>
__global__ void form_b(unsigned long long *out, double *io, int passes) {
  const int lane = threadIdx.x, row = lane >> 4, lr = lane & 15;
  double Hc[3], Nc[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { Hc[k] = io[64 * k + lane]; Nc[k] = io[64 * (12 + k) + lane]; }
  double npj = io[64 * 24 + lr]; // the robot's vectors are the same in every row
  const double zb = 0.0625;
  const int rot = ((lane & 48) | ((lr + 4 * row) & 15)) << 2; // lane k of row rho reads lane 4 rho + k
  unsigned long long t0, t1;
  STAMP(t0);
  const double npr0 = rotated(npj, rot);
  for (int p = 0; p < passes; p++) {
    const double npr = kHideNp ? npr0 : rotated(npj, rot);
    double zp = 0.0, rp = 0.0, zq = 0.0, rq = 0.0;
    fmac_bc<0, true>(zp, npr, Hc[0]); fmac_bc<0>(rp, npr, Nc[0]);
    fmac_bc<1>(zq, npr, Hc[1]);       fmac_bc<1>(rq, npr, Nc[1]);
    fmac_bc<2>(zp, npr, Hc[2]);       fmac_bc<2>(rp, npr, Nc[2]);
    const double z = rows_sum(zp + zq), r = rows_sum(rp + rq);
    const double zn = row_sum(fma(z, npj, zb));
    const double zinv = rcp_nr(zn);
    const double vec = z * zinv, hc = -z, nc = -r;
    const double vr = rotated(vec, rot);
    fmac_bc<0, true>(Hc[0], vr, hc); fmac_bc<0>(Nc[0], vr, nc);
    fmac_bc<1>(Hc[1], vr, hc);       fmac_bc<1>(Nc[1], vr, nc);
    fmac_bc<2>(Hc[2], vr, hc);       fmac_bc<2>(Nc[2], vr, nc);
  }
  STAMP(t1);
  double acc = npj;
#pragma unroll
  for (int k = 0; k < 3; k++) acc += Hc[k] + Nc[k];
  io[64 * 25 + lane] = acc;
  if (lane == 0) out[0] = t1 - t0;
}

int main() {
  const int passes = 200;
  unsigned long long *out;
  double *io;
  hipMalloc(&out, 64);
  hipMalloc(&io, 64 * 26 * sizeof(double));
  std::vector<double> h(64 * 26);
  // a positive definite H (identity plus a little), small N*, a unit-size n_p: the updates stay finite over 200 passes
  for (int j = 0; j < 12; j++)
    for (int l = 0; l < 64; l++) {
      h[64 * j + l] = ((l & 15) == lane_of(j) ? 1.0 : 0.0) + 1e-3 * ((j + l) % 7);
      h[64 * (12 + j) + l] = 1e-3 * ((3 * j + l) % 5);
    }
  for (int l = 0; l < 64; l++) h[64 * 24 + l] = ((l & 15) % 4 == 3) ? 0.0 : 1e-2 * (1 + (l & 15) % 3);
  double cyc[3][5];
  for (int rep = 0; rep < 5; rep++) {
    for (int form = 0; form < 3; form++) {
      hipMemcpy(io, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
      if (form == 0) hipLaunchKernelGGL(form_a, dim3(1), dim3(64), 0, 0, out, io, passes);
      else if (form == 1) hipLaunchKernelGGL(form_b<false>, dim3(1), dim3(64), 0, 0, out, io, passes);
      else hipLaunchKernelGGL(form_b<true>, dim3(1), dim3(64), 0, 0, out, io, passes);
      unsigned long long t = 0;
      hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
      cyc[form][rep] = (double)t / passes; // s_memtime ticks at the shader clock here (tools/stamp_probe_pose.py: 2.408 GHz)
    }
  }
  for (int form = 0; form < 3; form++) {
    double best = cyc[form][0];
    for (int rep = 1; rep < 5; rep++) best = cyc[form][rep] < best ? cyc[form][rep] : best;
    printf("form %s: %.0f cycles per pass (best of 5 launches of %d passes)\n",
           form == 0 ? "A (row per lane, 48 broadcast-FMAs)" : form == 1 ? "B (64 lanes, 12 broadcast-FMAs + exchange)"
                                                                       : "B, rotation of n_p not counted", best, passes);
  }
  return 0;
}
