// FP64 peak on gfx950, measured: v_fma_f64 (vector) against v_mfma_f64_16x16x4_f64 (matrix core), with 1, 2 and 4
// waves per SIMD.  Answers whether the D = 168 covariance products would run faster on MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_bench.hip -o tools/mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

// 16 independent FMA chains per lane: 2 flops x 64 lanes per instruction
__global__ void k_fma(double* out, int iters) {
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = threadIdx.x * 1e-3 + j;
  const double b = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = __builtin_fma(a[j], b, c);
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += a[j];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 8 independent accumulator tiles per wave: one v_mfma_f64_16x16x4_f64 = 16*16*4*2 = 2048 flops
__global__ void k_mfma(double* out, int iters) {
  double4_t acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double* d;
  hipMalloc(&d, (size_t)256 * 4 * 4 * 64 * 8 * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 100000;
  for (int wps : {1, 2, 4}) {  // waves per SIMD: 256 CUs x 4 SIMDs x wps waves, one workgroup of 4*wps waves per CU
    const int threads = 64 * 4 * wps, blocks = 256;
    for (int which = 0; which < 2; ++which) {
      auto launch = [&](int it) {
        if (which == 0) k_fma<<<blocks, threads>>>(d, it);
        else k_mfma<<<blocks, threads>>>(d, it);
      };
      launch(1000);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch(iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double waves = (double)blocks * threads / 64;
      const double flops = which == 0 ? waves * iters * 16.0 * 2.0 * 64.0 : waves * iters * 8.0 * 2048.0;
      const double inst = which == 0 ? iters * 16.0 : iters * 8.0;  // per wave
      // a SIMD runs wps waves: cycles per instruction per SIMD at the 2.4 GHz peak clock (upper bound; the clock drops under FP64 load)
      printf("%-22s %d wave(s)/SIMD: %8.3f ms  %7.2f TFLOP/s  %.1f cycles/instr/SIMD @2.4GHz\n",
             which == 0 ? "v_fma_f64" : "v_mfma_f64_16x16x4_f64", wps, ms, flops / (ms * 1e-3) / 1e12,
             ms * 1e-3 * 2.4e9 / (inst * wps));
    }
  }
  return 0;
}
