// Where does a step of the D = 168 matrix-core smoother (csrc/smooth_mfma.h) spend its time?  Diagnostic build with
// wall-clock stamps at the phase boundaries of workgroup 0, run on a synthetic set of filter records (random SPD
// covariances of Pleiades size) with `nwg` workgroups active, so that the memory system is loaded as in production.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -DODEF_MFMA_STAMPS -I odefilters.jl_amd/csrc tools/mfma_smooth_stamps.hip -o tools/mfma_smooth_stamps
#include "smooth_mfma.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <random>
using namespace odef;
constexpr int d = 28, q = 5, NB = 6, D = 168, TRI = D * (D + 1) / 2;
#ifndef WGS
#define WGS 4
#endif
__global__ __launch_bounds__(256, WGS) void kern(const SmoothParams P, double* ws) {
  using W = MfmaSmoothWs<d, NB>;
  __shared__ double lds[W::lds_size];
  smooth_mfma_traj<d, q>(P, (long)blockIdx.x, ws + (size_t)blockIdx.x * W::size, lds);
}
int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 512, ns = 8;
  const bool staged = argc > 2 && argv[2][0] == 's';  // records trajectory-major (record_stage.h), as odef_smooth runs fixed grids
  std::mt19937_64 rng(3);
  std::normal_distribution<double> nd;
  // one SPD covariance (scaled like a preconditioned-then-unpreconditioned state), reused for all records
  std::vector<double> F(D * 8), cov1(TRI);
  for (auto& x : F) x = nd(rng);
  for (int a = 0; a < D; ++a)
    for (int b = 0; b <= a; ++b) {
      double s = (a == b) ? 1.0 : 0.0;
      for (int k = 0; k < 8; ++k) s += F[a * 8 + k] * F[b * 8 + k];
      cov1[a * (a + 1) / 2 + b] = 1e-6 * s;
    }
  std::vector<double> cov((size_t)ns * TRI * N), mean((size_t)ns * D * N, 0.5), diff((size_t)ns * N, 1.0), hs(ns, 1.0 / 1024), ptab(kTabStride);
  for (long s = 0; s < ns; ++s)
    for (int e = 0; e < TRI; ++e)
      for (long i = 0; i < N; ++i) cov[((size_t)s * TRI + e) * N + i] = cov1[e];
  precond_fill<NB>(hs[0], std::pow(hs[0], -q - 0.5), ptab.data());
  std::vector<int> idx(ns, 0);
  SmoothParams P;
  std::memset(&P, 0, sizeof P);
  // prior tables: A = 1/(j-i)!, Q as in src/priors.jl
  for (int J = 0; J < NB; ++J) {
    double v = 1.0;
    for (int j = J; j < NB; ++j) { P.pc.At[J][j] = v; v /= (j - J + 1); }
    for (int K = 0; K < NB; ++K) {
      double f1 = 1, f2 = 1;
      for (int k = 2; k <= q - J; ++k) f1 *= k;
      for (int k = 2; k <= q - K; ++k) f2 *= k;
      P.pc.Qt[J][K] = 1.0 / ((2 * q + 1 - J - K) * f1 * f2);
    }
  }
  auto dev = [](const void* h, size_t b) { void* p; hipMalloc(&p, b); hipMemcpy(p, h, b, hipMemcpyHostToDevice); return p; };
  P.N = N; P.n_save = ns; P.adaptive = 0;
  P.ptab = (const double*)dev(ptab.data(), ptab.size() * 8); P.tab_idx = (const int*)dev(idx.data(), idx.size() * 4);
  P.hs = (const double*)dev(hs.data(), hs.size() * 8);
  P.mean = (const double*)dev(mean.data(), mean.size() * 8); P.cov = (const double*)dev(cov.data(), cov.size() * 8);
  P.diff = (const double*)dev(diff.data(), diff.size() * 8);
  void *sm, *sc, *rc, *ws;
  hipMalloc(&sm, mean.size() * 8); hipMalloc(&sc, cov.size() * 8); hipMalloc(&rc, N * 4); hipMemset(rc, 0, N * 4);
  hipMalloc(&ws, (size_t)N * MfmaSmoothWs<d, NB>::size * 8);
  P.smean = (double*)sm; P.scov = (double*)sc; P.retcode = (int*)rc;
  if (staged) {  // every record holds the same covariance: fill the stage directly
    const long ld = (TRI + 15) / 16 * 16;
    std::vector<double> st((size_t)(ns - 1) * N * ld, 0.0);
    for (size_t r = 0; r < (size_t)(ns - 1) * N; ++r)
      for (int e = 0; e < TRI; ++e) st[r * ld + e] = cov1[e];
    P.stage = (double*)dev(st.data(), st.size() * 8);
    P.stage_s0 = 1; P.stage_ld = ld; P.stage_hi = ns - 1;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    unsigned long long z[16] = {0};
#ifdef ODEF_MFMA_STAMPS
    hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_stamps), z, sizeof z);
#endif
    hipEventRecord(e0);
    if (staged) hipMemcpy(P.stage, P.stage, 0, hipMemcpyDeviceToDevice);
    kern<<<(unsigned)N, 256>>>(P, (double*)ws);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); return 1; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[16];
    for (auto& x : st) x = 1;
#ifdef ODEF_MFMA_STAMPS
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_mfma_stamps), sizeof st);
#endif
    const char* names[] = {"unpack X", "Yt = A X", "B, M", "Cholesky", "sweeps", "mean", "Z = M Gt", "R = Z' Gt", "pack + store"};
    double tot = 0;
    for (int k = 0; k < 9; ++k) tot += (double)st[k];
    printf("N = %ld workgroups, %ld steps each: %.2f ms total = %.0f us per step per workgroup-slot; workgroup 0 per step (100 MHz wall clock):\n", N, ns - 2, ms,
           ms * 1e3 / ((ns - 2) * ((N + 255) / 256)));
    for (int k = 0; k < 9; ++k) printf("  %-13s %8.1f us  %4.1f %%\n", names[k], st[k] / 100.0 / (ns - 2), 100.0 * st[k] / tot);
  }
  return 0;
}
