// Does a gfx950 FP64 VALU instruction with only 16 (or 32) active lanes issue faster than with 64?
// Build: hipcc --offload-arch=gfx950 -O3 tools/exec_mask_bench.hip -o tools/exec_mask_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(double* out, int active, int iters) {
  if ((int)threadIdx.x >= active) return;
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double b = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
    a0 = a0 * b + c; a1 = a1 * b + c; a2 = a2 * b + c; a3 = a3 * b + c;
    a4 = a4 * b + c; a5 = a5 * b + c; a6 = a6 * b + c; a7 = a7 * b + c;
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  double* d;
  hipMalloc(&d, 1024 * 64 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int active : {64, 32, 16, 8, 1}) {
    k<<<1024, 64>>>(d, active, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<1024, 64>>>(d, active, 200000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("active lanes %2d: %.3f ms  (%.2f cycles per FMA-instruction per wave at 2.4 GHz)\n", active, ms, ms * 1e-3 * 2.4e9 / (200000.0 * 8));
  }
  return 0;
}
