// Device-side unit test of csrc/mfma_dense.h: blocked Cholesky (upper, in place), the two block sweeps and the A'B
// product on the matrix cores, one workgroup of 4 wavefronts, against double-loop host references.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I odefilters.jl_amd/csrc tools/mfma_dense_test.hip -o tools/mfma_dense_test
#include "mfma_dense.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using namespace odef;
constexpr int D = 168, DPB = 11, DP = DPB * 16;

__global__ __launch_bounds__(256) void k_test(double* Bm, double* Lm, double* Yt, const double* M, double* Z2, double* R) {
  __shared__ double lds[mf::CholLds<DPB>::size];
  mf::wg_cholesky_upper<DPB>(Bm, Lm, DP, lds);
#ifndef TEST_RR  // the forms the smoother uses (block rows, a barrier after each; 3 x 2 tile strips)
  mf::wg_solve_upper<DPB>(Bm, Lm, Yt, DP, lds);
  __syncthreads();
  mf::wg_atb<false>(M, DP, Yt, DP, DP, nullptr, Z2, DP, 0, DPB, 0, DPB);   // Z2 = M' Gt
  __syncthreads();
  mf::wg_atb<false>(Z2, DP, Yt, DP, DP, nullptr, R, DP, 0, DPB, 0, DPB);   // R = Z2' Gt
#else            // the register-resident variants (build with -DTEST_RR): same results, slower
  mf::wg_solve_upper_rr<DPB>(Bm, Lm, Yt, DP, lds);
  __syncthreads();
  mf::wg_atb_rescols<DPB>(M, Yt, Z2, DP);   // Z2 = M' Gt
  __syncthreads();
  mf::wg_atb_rescols<DPB>(Z2, Yt, R, DP);   // R = Z2' Gt
#endif
}

int main() {
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  std::vector<double> F(DP * DP, 0.0), B(DP * DP, 0.0), Y(DP * DP, 0.0), M(DP * DP, 0.0);
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) F[i * DP + j] = nd(rng) / std::sqrt((double)D);
  for (int i = 0; i < D; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = (i == j) ? 0.5 : 0.0;
      for (int k = 0; k < D; ++k) s += F[i * DP + k] * F[j * DP + k];
      B[i * DP + j] = B[j * DP + i] = s;
      const double m = nd(rng);
      M[i * DP + j] = M[j * DP + i] = m;
    }
  for (int i = D; i < DP; ++i) B[i * DP + i] = 1.0;
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) Y[i * DP + j] = nd(rng);
  // host reference: Gt = B^-1 Yt by plain Cholesky
  std::vector<double> L(B), G(Y);
  for (int k = 0; k < DP; ++k) {
    L[k * DP + k] = std::sqrt(L[k * DP + k]);
    for (int i = k + 1; i < DP; ++i) L[i * DP + k] /= L[k * DP + k];
    for (int j = k + 1; j < DP; ++j)
      for (int i = j; i < DP; ++i) L[i * DP + j] -= L[i * DP + k] * L[j * DP + k];
  }
  for (int c = 0; c < DP; ++c) {
    for (int i = 0; i < DP; ++i) {
      double t = G[i * DP + c];
      for (int k = 0; k < i; ++k) t -= L[i * DP + k] * G[k * DP + c];
      G[i * DP + c] = t / L[i * DP + i];
    }
    for (int i = DP - 1; i >= 0; --i) {
      double t = G[i * DP + c];
      for (int k = i + 1; k < DP; ++k) t -= L[k * DP + i] * G[k * DP + c];
      G[i * DP + c] = t / L[i * DP + i];
    }
  }
  std::vector<double> Z2r(DP * DP, 0.0), Rr(DP * DP, 0.0);
  for (int a = 0; a < DP; ++a)
    for (int c = 0; c < DP; ++c) {
      double s = 0;
      for (int k = 0; k < DP; ++k) s += M[k * DP + a] * G[k * DP + c];
      Z2r[a * DP + c] = s;
    }
  for (int a = 0; a < DP; ++a)
    for (int c = 0; c < DP; ++c) {
      double s = 0;
      for (int k = 0; k < DP; ++k) s += Z2r[k * DP + a] * G[k * DP + c];
      Rr[a * DP + c] = s;
    }
  double *dB, *dL, *dY, *dM, *dZ, *dR;
  const size_t nb = sizeof(double) * DP * DP;
  hipMalloc(&dB, nb); hipMalloc(&dL, nb); hipMalloc(&dY, nb); hipMalloc(&dM, nb); hipMalloc(&dZ, nb); hipMalloc(&dR, nb);
  hipMemcpy(dB, B.data(), nb, hipMemcpyHostToDevice); hipMemcpy(dY, Y.data(), nb, hipMemcpyHostToDevice);
  hipMemcpy(dM, M.data(), nb, hipMemcpyHostToDevice); hipMemset(dL, 0, nb);
  k_test<<<1, 256>>>(dB, dL, dY, dM, dZ, dR);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  std::vector<double> U(DP * DP), Gd(DP * DP), Rd(DP * DP), Ld(DP * DP);
  hipMemcpy(U.data(), dB, nb, hipMemcpyDeviceToHost); hipMemcpy(Gd.data(), dY, nb, hipMemcpyDeviceToHost);
  hipMemcpy(Rd.data(), dR, nb, hipMemcpyDeviceToHost); hipMemcpy(Ld.data(), dL, nb, hipMemcpyDeviceToHost);
  double eu = 0, el = 0, eg = 0, er = 0, sg = 0, sr = 0;
  for (int i = 0; i < DP; ++i)
    for (int j = i; j < DP; ++j) {
      eu = std::fmax(eu, std::fabs(U[i * DP + j] - L[j * DP + i]));
      el = std::fmax(el, std::fabs(Ld[j * DP + i] - L[j * DP + i]));
    }
  for (int i = 0; i < DP * DP; ++i) {
    eg = std::fmax(eg, std::fabs(Gd[i] - G[i])); sg = std::fmax(sg, std::fabs(G[i]));
    er = std::fmax(er, std::fabs(Rd[i] - Rr[i])); sr = std::fmax(sr, std::fabs(Rr[i]));
  }
  printf("U vs chol' %.2e   L copy %.2e   Gt rel %.2e   R = G M G' rel %.2e\n", eu, el, eg / sg, er / sr);
  const bool ok = eu < 1e-11 && el < 1e-11 && eg / sg < 1e-10 && er / sr < 1e-10;
  printf("%s\n", ok ? "mfma dense ok" : "mfma dense MISMATCH");
  // timing: 200 repetitions of the whole chain on 256 workgroups would need per-workgroup matrices; one workgroup here
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemcpy(dB, B.data(), nb, hipMemcpyHostToDevice); hipMemcpy(dY, Y.data(), nb, hipMemcpyHostToDevice);
  hipEventRecord(e0);
  k_test<<<1, 256>>>(dB, dL, dY, dM, dZ, dR);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("one workgroup, Cholesky + sweeps + two products: %.1f us\n", ms * 1e3);
  return ok ? 0 : 1;
}
