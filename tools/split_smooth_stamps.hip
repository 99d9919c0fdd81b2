// Where does a record of the D = 168 smoother's split pass spend its time?  Diagnostic build with wall-clock stamps at the
// phase boundaries of workgroup 0 of the on-chip kernel (rts_smooth_sweeps_kernel, csrc/ek_kernels.h), run on a synthetic
// set of staged filter records (random SPD covariances of Pleiades size) with N trajectories, and the time per launch of
// both kernels of the pass (rts_smooth_predict_kernel, rts_smooth_sweeps_kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -DODEF_SWEEPS_STAMPS -I odefilters.jl_amd/csrc tools/split_smooth_stamps.hip -o tools/_bin/split_smooth_stamps
#include "ek_kernels.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <random>
namespace odef {  // (the library defines these in api.hip)
static thread_local char g_kn[256];
void note_kernel(const char*, ...) {}
const char* last_kernel() { return g_kn; }
}
using namespace odef;
constexpr int d = 28, q = 5, NB = 6, D = 168, TRI = D * (D + 1) / 2;
int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 2048, ns = 10;
  std::mt19937_64 rng(3);
  std::normal_distribution<double> nd;
  std::vector<double> F(D * 8), cov1(TRI);
  for (auto& x : F) x = nd(rng);
  for (int a = 0; a < D; ++a)
    for (int b = 0; b <= a; ++b) {
      double s = (a == b) ? 1.0 : 0.0;
      for (int k = 0; k < 8; ++k) s += F[a * 8 + k] * F[b * 8 + k];
      cov1[a * (a + 1) / 2 + b] = 1e-6 * s;
    }
  std::vector<double> mean((size_t)ns * D * N, 0.5), diff((size_t)ns * N, 1.0), hs(ns, 1.0 / 1024), ptab(kTabStride);
  precond_fill<NB>(hs[0], std::pow(hs[0], -q - 0.5), ptab.data());
  std::vector<int> idx(ns, 0);
  SmoothParams P;
  std::memset(&P, 0, sizeof P);
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
  auto dev = [](const void* h, size_t b) { void* p; (void)hipMalloc(&p, b); (void)hipMemcpy(p, h, b, hipMemcpyHostToDevice); return p; };
  P.N = N; P.n_save = ns; P.adaptive = 0;
  P.ptab = (const double*)dev(ptab.data(), ptab.size() * 8); P.tab_idx = (const int*)dev(idx.data(), idx.size() * 4);
  P.hs = (const double*)dev(hs.data(), hs.size() * 8);
  P.mean = (const double*)dev(mean.data(), mean.size() * 8);
  P.diff = (const double*)dev(diff.data(), diff.size() * 8);
  void *sm, *rc, *ws;
  (void)hipMalloc(&sm, mean.size() * 8); (void)hipMalloc(&rc, N * 4); (void)hipMemset(rc, 0, N * 4);
  (void)hipMalloc(&ws, (size_t)N * MfmaSmoothWs<d, NB>::size * 8);
  P.smean = (double*)sm; P.retcode = (int*)rc;
  const long ld = (TRI + 15) / 16 * 16;
  {
    std::vector<double> st((size_t)(ns - 1) * N * ld, 0.0);
    for (size_t r = 0; r < (size_t)(ns - 1) * N; ++r)
      for (int e = 0; e < TRI; ++e) st[r * ld + e] = cov1[e];
    P.stage = (double*)dev(st.data(), st.size() * 8);
    P.stage_s0 = 1; P.stage_ld = ld; P.stage_hi = ns - 1;
  }
  hipStream_t s = nullptr;
  hipEvent_t ev[64];
  for (auto& e : ev) (void)hipEventCreate(&e);
  for (int rep = 0; rep < 2; ++rep) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sweeps_stamps), z, sizeof z);
    P.split_mode = 1; P.split_sc = P.split_sa = -1;
    { LaunchTeamSmooth f{P, (double*)ws, s}; f.operator()<d, q>(); }
    P.split_mode = 2;
    P.split_sc = (argc > 2 && argv[2][0] == 'w') ? -1 : 1;  // 'w': Y' through the workspace (as before), default: formed on chip
    int ne = 0;
    for (long r = ns - 2; r >= 1; --r) {
      P.split_sa = r;
      (void)hipEventRecord(ev[ne++], s);
      { LaunchTeamSmoothPredict f{P, (double*)ws, s}; f.operator()<d, q>(); if (f.rc) { printf("launch failed\n"); return 1; } }
      (void)hipEventRecord(ev[ne++], s);
      LaunchTeamSmoothSweeps g{P, (double*)ws, s};
      g.operator()<d, q>();
      if (g.rc) { printf("launch failed\n"); return 1; }
    }
    (void)hipEventRecord(ev[ne++], s);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    double k1 = 0, k2 = 0;
    for (int e = 0; e + 2 < ne + 1 && e + 2 <= ne - 1 + 1; e += 2) {
      float a, b;
      (void)hipEventElapsedTime(&a, ev[e], ev[e + 1]);
      (void)hipEventElapsedTime(&b, ev[e + 1], ev[e + 2]);
      k1 += a; k2 += b;
    }
    const int nrec = (int)(ns - 2);
    unsigned long long st[16];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_sweeps_stamps), sizeof st);
    const char* names[] = {"B -> LDS", "factorisation", "Y -> acc", "forward", "backward", "M, vec -> LDS", "mean", "G M G'", "X + R, stores"};
    double tot = 0;
    for (int k = 0; k < 9; ++k) tot += (double)st[k];
    printf("N = %ld, %d records: unpack/predict kernel %.3f ms, on-chip kernel %.3f ms per record; on-chip kernel, workgroup 0 per record (100 MHz wall clock), sum %.1f us:\n", N, nrec,
           k1 / nrec, k2 / nrec, tot / 100.0 / nrec);
    for (int k = 0; k < 9; ++k) printf("  %-15s %8.1f us  %4.1f %%\n", names[k], st[k] / 100.0 / nrec, 100.0 * st[k] / tot);
  }
  return 0;
}
