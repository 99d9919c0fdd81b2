// FP64 FMA on gfx950, one wavefront per SIMD (1 024 workgroups of 64): cycles per instruction for CHAINS independent
// dependency chains (1 = fully dependent) and for 64 / 32 / 16 active lanes.  What a lane-per-trajectory kernel with one
// wavefront per SIMD can hope for.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fma_latency_bench.hip -o tools/fma_latency_bench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ __launch_bounds__(64) void k(double* out, int active, int iters) {
  if ((int)threadIdx.x >= active) return;
  double a[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) a[c] = threadIdx.x * 1e-3 + c;
  const double b = 1.0000001, cc = 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8 / CHAINS; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) a[c] = a[c] * b + cc;
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += a[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CHAINS>
void run(double* d, int wgs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int active : {64, 32, 16}) {
    k<CHAINS><<<wgs, 64>>>(d, active, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAINS><<<wgs, 64>>>(d, active, 200000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD %d chains %d active lanes %2d: %.3f ms = %.2f ns per FMA instruction per wave\n", wgs / 1024, CHAINS, active, ms, ms * 1e6 / (200000.0 * 8));
  }
}
int main() {
  double* d;
  hipMalloc(&d, 4096 * 64 * 8);
  for (int wgs : {1024, 2048}) {
    run<1>(d, wgs); run<2>(d, wgs); run<4>(d, wgs); run<8>(d, wgs);
  }
  return 0;
}
