// Accuracy of v_rcp_f64 / v_rsq_f64 seeds and of one / two Newton steps (diagnostic).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y0 = __builtin_amdgcn_rcp(v);
  double y1 = y0 + y0 * (1.0 - v * y0);
  double y1b = fma(y0, fma(-v, y0, 1.0), y0);
  double y2 = y1b + y1b * (1.0 - v * y1b);
  double r0 = __builtin_amdgcn_rsq(v);
  double r1 = r0 + r0 * (0.5 - 0.5 * v * r0 * r0);
  out[6 * i] = y0; out[6 * i + 1] = y1; out[6 * i + 2] = y1b; out[6 * i + 3] = y2; out[6 * i + 4] = r0; out[6 * i + 5] = r1;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), o(6 * n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = std::ldexp(1.0 + (s >> 11) * (1.0 / 9007199254740992.0), (int)(s % 60) - 30); }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
  hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
  double e[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    const long double t = 1.0L / (long double)h[i], q = 1.0L / sqrtl((long double)h[i]);
    for (int k2 = 0; k2 < 4; k2++) { double r = (double)fabsl(((long double)o[6 * i + k2] - t) / t); if (r > e[k2]) e[k2] = r; }
    for (int k2 = 4; k2 < 6; k2++) { double r = (double)fabsl(((long double)o[6 * i + k2] - q) / q); if (r > e[k2]) e[k2] = r; }
  }
  printf("max relative error: rcp seed %.3e  one step %.3e  one step (fma form) %.3e  two steps %.3e | rsq seed %.3e  one step %.3e\n", e[0], e[1], e[2], e[3], e[4], e[5]);
  return 0;
}
