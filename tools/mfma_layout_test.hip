// Operand layout of v_mfma_f64_16x16x4_f64 on gfx950, checked against a host product: D(16x16) = A(16x4) B(4x16) + C with
//   A: lane l holds A[i = l % 16][k = l / 16]            (one double)
//   B: lane l holds B[k = l / 16][j = l % 16]            (one double)
//   C/D: lane l holds D[i = 4 v + l / 16][j = l % 16], v = 0..3     (four doubles; NOT 4 (l / 16) + v)
// Prints "layout ok" when the assumption holds.  Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_layout_test.hip -o tools/mfma_layout_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];
  const double b = B[(l / 16) * 16 + l % 16];
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[(4 * v + l / 16) * 16 + l % 16] = acc[v];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + 0.37 * i; hB[i] = -2.0 + 0.11 * i * i; }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j];
      ref[i * 16 + j] = s;
    }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0; double worst = 0;
  for (int i = 0; i < 256; ++i) { double e = hD[i] - ref[i]; if (e < 0) e = -e; if (e > 1e-9 * (1 + (ref[i] < 0 ? -ref[i] : ref[i]))) ++bad; if (e > worst) worst = e; }
  printf("%s (%d mismatches, max abs err %.3e)\n", bad ? "layout MISMATCH" : "layout ok", bad, worst);
  return bad != 0;
}
