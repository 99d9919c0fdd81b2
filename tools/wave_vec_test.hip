// Device check of wave_vec.h (wavefront-register Cholesky quad-form and Householder QR) against plain host loops.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I odefilters.jl_amd/csrc tools/wave_vec_test.hip -o tools/wave_vec_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wave_vec.h"
using namespace odef;
constexpr int d = 28, d2 = 56, LD = 29;

__global__ __launch_bounds__(128) void k_test(const double* Win, const double* zin, const double* Gin, double* out_acc, double* outHV,
                       double* outbeta, double* outR, double* ms) {
  __shared__ double WM[d * LD], z[d], G[d2 * d], HV[d * d2], beta[d], R[d * d];
  for (int e = threadIdx.x; e < d * LD; e += blockDim.x) WM[e] = Win[e];
  for (int e = threadIdx.x; e < d; e += blockDim.x) z[e] = zin[e];
  for (int e = threadIdx.x; e < d2 * d; e += blockDim.x) G[e] = Gin[e];
  for (int e = threadIdx.x; e < d * d2; e += blockDim.x) HV[e] = 0.0;
  for (int e = threadIdx.x; e < d * d; e += blockDim.x) R[e] = 0.0;
  __syncthreads();
  if (threadIdx.x < 64) {
    // MultiSum self-test: value j of lane l is (l + 1) * (j + 1)
    double p[32];
    for (int j = 0; j < 32; ++j) p[j] = (threadIdx.x + 1.0) * (j + 1.0);
    wv::MultiSum<32>::run(p);
    for (int j = 0; j < 32; ++j) {
      const double r = wv::bcast(p[0], wv::MultiSum<32>::owner(j));
      if (threadIdx.x == 0) ms[j] = r;
    }
    const double acc = wv::chol_quadform<d>(wv::lds(WM), LD, wv::lds(z));
    if (threadIdx.x == 0) out_acc[0] = acc;
    __shared__ double yv[d];
    double zSz, logacc;
    wv::householder_qr_solve<d>(wv::lds(G), wv::lds(z), wv::lds(HV), wv::lds(beta), wv::lds(yv), zSz, logacc, wv::lds(R));
    if (threadIdx.x == 0) { out_acc[1] = zSz; out_acc[2] = logacc; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < d * d2; e += blockDim.x) outHV[e] = HV[e];
  for (int e = threadIdx.x; e < d; e += blockDim.x) outbeta[e] = beta[e];
  for (int e = threadIdx.x; e < d * d; e += blockDim.x) outR[e] = R[e];
}

int main(int argc, char** argv) {
  std::vector<double> W(d * LD, 0.0), z(d), G(d2 * d), M(d * d);
  srand(1);
  auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
  for (auto& v : M) v = rnd();
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      double s = (i == j) ? 0.5 : 0.0;
      for (int k = 0; k < d; ++k) s += M[i * d + k] * M[j * d + k];
      W[i * LD + j] = s;
    }
  for (auto& v : z) v = rnd();
  for (auto& v : G) v = rnd();
  if (argc > 1) { FILE* f = fopen(argv[1], "rb"); if (!f || fread(G.data(), 8, G.size(), f) != G.size()) { printf("cannot read %s\n", argv[1]); return 2; } fclose(f); }
  // host references
  std::vector<double> L(d * d, 0.0), y(d);
  {
    std::vector<double> A(d * d);
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) A[i * d + j] = W[i * LD + j];
    for (int k = 0; k < d; ++k) {
      L[k * d + k] = sqrt(A[k * d + k]);
      for (int i = k + 1; i < d; ++i) L[i * d + k] = A[i * d + k] / L[k * d + k];
      for (int i = k + 1; i < d; ++i) for (int j = k + 1; j <= i; ++j) A[i * d + j] -= L[i * d + k] * L[j * d + k];
    }
  }
  double acc_ref = 0.0;
  for (int r = 0; r < d; ++r) { double s = z[r]; for (int c = 0; c < r; ++c) s -= L[r * d + c] * y[c]; y[r] = s / L[r * d + r]; acc_ref += y[r] * y[r]; }
  // R'R = G'G check
  std::vector<double> GtG(d * d, 0.0);
  for (int a = 0; a < d; ++a) for (int b = 0; b < d; ++b) for (int i = 0; i < d2; ++i) GtG[a * d + b] += G[i * d + a] * G[i * d + b];
  double *dW, *dz, *dG, *dacc, *dHV, *dbeta, *dR, *dms;
  hipMalloc(&dW, W.size() * 8); hipMalloc(&dz, z.size() * 8); hipMalloc(&dG, G.size() * 8); hipMalloc(&dacc, 24);
  hipMalloc(&dHV, d * d2 * 8); hipMalloc(&dbeta, d * 8); hipMalloc(&dR, d * d * 8); hipMalloc(&dms, 32 * 8);
  hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dz, z.data(), z.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice);
  k_test<<<1, 128>>>(dW, dz, dG, dacc, dHV, dbeta, dR, dms);
  double acc3[3], ms[32];
  std::vector<double> HV(d * d2), beta(d), R(d * d);
  hipMemcpy(acc3, dacc, 24, hipMemcpyDeviceToHost);
  const double acc = acc3[0];
  hipMemcpy(ms, dms, 32 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(HV.data(), dHV, HV.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(beta.data(), dbeta, d * 8, hipMemcpyDeviceToHost);
  hipMemcpy(R.data(), dR, R.size() * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int j = 0; j < 32; ++j) { const double want = 2080.0 * (j + 1); if (ms[j] != want) { printf("multisum[%d] = %g want %g\n", j, ms[j], want); ++bad; } }
  printf("quadform dev %.17g ref %.17g rel %.2e\n", acc, acc_ref, fabs(acc - acc_ref) / acc_ref);
  if (!(fabs(acc - acc_ref) <= 1e-12 * acc_ref)) ++bad;
  double worst = 0.0;
  for (int a = 0; a < d; ++a) for (int b = 0; b < d; ++b) {
    double s = 0.0;
    for (int k = 0; k <= (a < b ? a : b); ++k) s += R[k * d + a] * R[k * d + b];
    worst = fmax(worst, fabs(s - GtG[a * d + b]));
  }
  double gmax = 0.0; for (double v : GtG) gmax = fmax(gmax, fabs(v));
  worst /= gmax;
  printf("max |R'R - G'G| / max|G'G| = %.2e\n", worst);
  if (!(worst < 1e-12)) ++bad;
  // apply reflectors to G: must give R on top, zeros below
  std::vector<double> T = G;
  for (int k = 0; k < d; ++k) for (int c = 0; c < d; ++c) {
    double s = 0.0;
    for (int i = k; i < d2; ++i) s += HV[k * d2 + i] * T[i * d + c];
    s *= beta[k];
    for (int i = k; i < d2; ++i) T[i * d + c] -= s * HV[k * d2 + i];
  }
  double w2 = 0.0;
  for (int i = 0; i < d2; ++i) for (int c = 0; c < d; ++c) w2 = fmax(w2, fabs(T[i * d + c] - ((i <= c) ? R[i * d + c] : 0.0)));
  double gm = 0.0; for (double v : G) gm = fmax(gm, fabs(v));
  w2 /= gm;
  printf("max |Q'G - R| / max|G| = %.2e\n", w2);
  if (!(w2 < 1e-12)) ++bad;
  {  // y = R^-T z: z'S^-1 z and log det from the host's R
    double yh[d], zs = 0.0, la = 0.0;
    for (int r = 0; r < d; ++r) { double s = z[r]; for (int c = 0; c < r; ++c) s -= R[c * d + r] * yh[c]; yh[r] = s / R[r * d + r]; zs += yh[r] * yh[r]; la += log(fabs(R[r * d + r])); }
    printf("zSz dev %.15g ref %.15g, logdet dev %.15g ref %.15g\n", acc3[1], zs, acc3[2], la);
    if (!(fabs(acc3[1] - zs) <= 1e-10 * fabs(zs)) || !(fabs(acc3[2] - la) <= 1e-10 * fabs(la) + 1e-12)) ++bad;
  }
  printf(bad ? "FAIL\n" : "OK\n");
  return bad ? 1 : 0;
}
