// Accuracy of the coupled sqrt / 1/sqrt (ek_math.h sqrt_and_rsqrt: v_rsq_f64 seed + two Goldschmidt steps) against the
// correctly rounded sqrt() and 1.0 / sqrt(): maximum error in ulps over 2^22 random positive doubles, exponents -300..300.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I odefilters.jl_amd/csrc tools/rsqrt_accuracy.hip -o tools/rsqrt_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "ek_math.h"
__global__ void k(const double* x, double* s, double* rs, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) odef::sqrt_and_rsqrt(x[i], s[i], rs[i]);
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), s(n), rs(n);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> m(1.0, 2.0);
  std::uniform_int_distribution<int> e(-300, 300);
  for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), e(g));
  double *dx, *ds, *dr;
  hipMalloc(&dx, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dr, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, ds, dr, n);
  hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(rs.data(), dr, n * 8, hipMemcpyDeviceToHost);
  double es = 0, er = 0;
  for (int i = 0; i < n; ++i) {
    const long double ts = sqrtl((long double)x[i]), tr = 1.0L / ts;
    const double us = std::ldexp(1.0, std::ilogb((double)ts) - 52), ur = std::ldexp(1.0, std::ilogb((double)tr) - 52);
    es = std::fmax(es, (double)(fabsl((long double)s[i] - ts) / us));
    er = std::fmax(er, (double)(fabsl((long double)rs[i] - tr) / ur));
  }
  printf("sqrt_and_rsqrt over %d samples: max error sqrt %.3f ulp, 1/sqrt %.3f ulp (correctly rounded = 0.5)\n", n, es, er);
  return 0;
}
