// DIAGNOSTIC BUILD ONLY: in-kernel cycle stamps of the MFMA Pleiades filter step
// (filter_mfma.h compiled with -DODEF_MF_STAMPS).  Prints the share of each segment of the step.
// Never quote this build's run time (the stamps serialise); read the shares.
#define ODEF_MF_STAMPS 1
#include "../odefilters.jl_amd/csrc/ek_kernels.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
using namespace odef;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static void build_prior(int q, PriorConsts& pc) {
  memset(&pc, 0, sizeof pc);
  const int nb = q + 1;
  for (int J = 0; J < nb; ++J) pc.At[J][J] = 1.0;
  double val = 1.0;
  for (int i = 1; i <= q; ++i) { val /= i; for (int J = 0; J + i < nb; ++J) pc.At[J][J + i] = val; }
  auto fact = [](int n) { double f = 1; for (int k = 2; k <= n; ++k) f *= k; return f; };
  for (int c = 0; c < nb; ++c) for (int r = c; r < nb; ++r) { double v = 1.0 / ((2 * q + 1 - r - c) * fact(q - r) * fact(q - c)); pc.Qt[r][c] = pc.Qt[c][r] = v; }
  for (int j = 0; j < nb; ++j) {
    double s = pc.Qt[j][j]; for (int k = 0; k < j; ++k) s -= pc.QLt[j][k] * pc.QLt[j][k];
    pc.QLt[j][j] = sqrt(s);
    for (int i = j + 1; i < nb; ++i) { double t = pc.Qt[i][j]; for (int k = 0; k < j; ++k) t -= pc.QLt[i][k] * pc.QLt[j][k]; pc.QLt[i][j] = t / pc.QLt[j][j]; }
  }
}

int main() {
  constexpr int q = 5, d = 28, D = d * (q + 1), TRI = D * (D + 1) / 2;
  const long N = 256, nsteps = 16;
  FilterParams P; memset(&P, 0, sizeof P);
  build_prior(q, P.pc);
  const double u0h[28] = {3, 3, -1, -3, 2, -2, 2, 3, -3, 2, 0, 0, -4, 4, 0, 0, 0, 0, 0, 1.75, -1.5, 0, 0, 0, -1.25, 1, 0, 0};
  std::vector<double> u0(d * N); for (int a = 0; a < d; ++a) for (long i = 0; i < N; ++i) u0[a * N + i] = u0h[a] + 1e-4 * (i % 7);
  double *du0, *dtab, *dhs, *dmean, *dcov, *ddiff, *dll; int *didx, *di5; unsigned long long* dst;
  CK(hipMalloc(&du0, sizeof(double) * d * N)); CK(hipMemcpy(du0, u0.data(), sizeof(double) * d * N, hipMemcpyHostToDevice));
  std::vector<double> tab(kTabStride); const double h = 1.0 / 1024; precond_fill<q + 1>(h, pow(h, -q - 0.5), tab.data());
  CK(hipMalloc(&dtab, sizeof(double) * kTabStride)); CK(hipMemcpy(dtab, tab.data(), sizeof(double) * kTabStride, hipMemcpyHostToDevice));
  std::vector<int> idx(nsteps, 0); std::vector<double> hs(nsteps, h);
  CK(hipMalloc(&didx, sizeof(int) * nsteps)); CK(hipMemcpy(didx, idx.data(), sizeof(int) * nsteps, hipMemcpyHostToDevice));
  CK(hipMalloc(&dhs, sizeof(double) * nsteps)); CK(hipMemcpy(dhs, hs.data(), sizeof(double) * nsteps, hipMemcpyHostToDevice));
  CK(hipMalloc(&dmean, sizeof(double) * D * N)); CK(hipMalloc(&dcov, sizeof(double) * TRI * N)); CK(hipMalloc(&ddiff, sizeof(double) * N)); CK(hipMalloc(&dll, sizeof(double) * N));
  CK(hipMalloc(&di5, sizeof(int) * N * 6)); CK(hipMalloc(&dst, sizeof(unsigned long long) * 32)); CK(hipMemset(dst, 0, sizeof(unsigned long long) * 32));
  P.u0 = du0; P.p = nullptr; P.p_shared = 1; P.N = N; P.ptab = dtab; P.tab_idx = didx; P.hs = dhs; P.nsteps = nsteps; P.everystep = 0; P.want_loglik = 1;
  P.mean = dmean; P.cov = dcov; P.diff = ddiff; P.loglik = dll; P.naccept = di5; P.nreject = di5 + N; P.nf = di5 + 2 * N; P.njac = di5 + 3 * N; P.nsaved = di5 + 4 * N; P.retcode = di5 + 5 * N;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_mf_stamp_buf), &dst, sizeof(dst)));
  hipLaunchKernelGGL((ek_filter_mfma_kernel<RhsPleiades, q, true>), dim3((unsigned)N), dim3(kMfBlock), 0, 0, P);
  CK(hipDeviceSynchronize());
  unsigned long long st[32]; CK(hipMemcpy(st, dst, sizeof st, hipMemcpyDeviceToHost));
  const char* names[16] = {"(between steps / save)", "(unused)", "(unused)", "exchange put of the tiles", "congruence stage 1 | helper: chol(H Q H') first half",
                           "exchange put Z", "congruence stage 2 | helper: second half, sigma2", "copies of the first-column tiles", "C0 = (A S A') H'", "(unused)",
                           "helper: Sm = H C, Cholesky, W = L^-1 | tiles: + sigma2 Q", "V = C W', K = V W | helper: y, loglik", "mean, T = S^- - V V', tile copies", "(unused)",
                           "E = T H' | helper: next step's mean, f, J, z, H0, M0", "S = T - E K', un-precondition | helper: next W, Hs0"};
  double tot = 0; for (int k = 1; k < 16; ++k) tot += (double)st[k];
  printf("segment shares of one step (block 0 of %ld, mean over %ld steps, stamps by thread 0); %.0f cycles = %.1f us per step at 2.4 GHz\n", N, nsteps, tot / nsteps, tot / nsteps / 2400.0);
  for (int k = 1; k < 16; ++k) printf("  %-58s %6.2f %%   %8.0f cycles/step\n", names[k], 100.0 * st[k] / tot, (double)st[k] / nsteps);
  const char* hn[11] = {"(waiting / other)", "chol(H Q H') first half", "chol(H Q H') second half", "y = L^-1 z, sigma2", "Sm blocks (24 MFMAs)", "factor_s: Cholesky of Sm, W = L^-1",
                        "y = W z, loglik", "chain_a1: P m, A m, pairs, f, J", "chain_a2: z, H0, M0", "chain_b: W = M0 M0' (MFMA)", "chain_c: Hs0"};
  printf("the helper wavefront's own work items (its clock):\n");
  for (int k = 1; k < 11; ++k) printf("  %-58s %8.0f cycles/step\n", hn[k], (double)st[16 + k] / nsteps);
  return 0;
}
