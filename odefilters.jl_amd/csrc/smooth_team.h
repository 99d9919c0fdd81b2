// RTS smoother driver: one TEAM of threads per trajectory, matrices in a shared workspace
// (src/smoothing.jl:4-28 `smooth_all!`).  Algorithmic traffic per step: read the filter record,
// write the smoothed record (2*B_alg - 8 bytes); the filter covariance is read a second time for
// the final sum (served by L1/L2).
#pragma once
#include "ek_lane.h"
#include "team.h"

namespace odef {

template <int d, int q, int TEAM>
__device__ inline void smooth_team_lane(const SmoothParams& P, long i, int tid, double* __restrict__ ws) {
  constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  using W = SmoothWs<d, NB>;
  constexpr int LD = W::LD;
  const Team<TEAM> t{tid};
  const long n = P.adaptive ? (long)P.nsaved[i] : P.n_save;
  const size_t N = (size_t)P.N;
  double* X = ws + W::X;
  double* M = ws + W::M;
  double* mf = ws + W::MF;
  double* ms = ws + W::MS;
  // first and last record are copied (index 1 in Julia is never smoothed, src/smoothing.jl:11)
  for (int w = 0; w < 2; ++w) {
    const long s = w == 0 ? 0 : n - 1;
    ODEF_TEAM_FOR(k, D) {
      const double v = P.mean[((size_t)s * D + k) * N + i];
      P.smean[((size_t)s * D + k) * N + i] = v;
      ms[k] = v;
    }
    ODEF_TEAM_FOR(e, D * D) {
      const int a = e / D, b = e % D;
      if (a >= b) {
        const double v = P.cov[((size_t)s * TRI + tri(a, b)) * N + i];
        P.scov[((size_t)s * TRI + tri(a, b)) * N + i] = v;
        M[a * LD + b] = v;
        M[b * LD + a] = v;
      }
    }
  }
  t.sync();
  bool nan_seen = false;
  for (long s = n - 2; s >= 1; --s) {
    double h;
    double tabv[kTabStride];
    const double* tab;
    if (P.adaptive) {
      h = P.tsave[(size_t)(s + 1) * N + i] - P.tsave[(size_t)s * N + i];
      if (h != 0.0) precond_fill<NB>(h, precond_val<q>(h), tabv);
      tab = tabv;
    } else {
      h = P.hs[s];
      tab = P.ptab + (size_t)P.tab_idx[s] * kTabStride;
    }
    if (h == 0.0) {  // src/smoothing.jl:13-16
      ODEF_TEAM_FOR(k, D) P.smean[((size_t)s * D + k) * N + i] = ms[k];
      ODEF_TEAM_FOR(e, D * D) {
        const int a = e / D, b = e % D;
        if (a >= b) P.scov[((size_t)s * TRI + tri(a, b)) * N + i] = M[a * LD + b];
      }
      continue;
    }
    const double sigma2 = P.diff[(size_t)(s + 1) * N + i];
    ODEF_TEAM_FOR(k, D) mf[k] = P.mean[((size_t)s * D + k) * N + i];
    ODEF_TEAM_FOR(e, D * D) {
      const int a = e / D, b = e % D;
      if (a >= b) {
        const double v = P.cov[((size_t)s * TRI + tri(a, b)) * N + i];
        X[a * LD + b] = v;
        X[b * LD + a] = v;
      }
    }
    t.sync();
    team_smooth_step<d, NB, TEAM>(t, P.pc, tab, sigma2, ws);
    // Sigma^s = P^-1 (P Sigma P + G (Sigma^s_+ - Sigma^-) G') P^-1 ; store, and keep as the carried state
    ODEF_TEAM_FOR(e, D * D) {
      const int a = e / D, b = e % D;
      if (a >= b) {
        const int blk = (a / d) * MAXNB + (b / d);
        const double c = P.cov[((size_t)s * TRI + tri(a, b)) * N + i];
        const double v = (c * tab[kTabPP + blk] + M[a * LD + b]) * tab[kTabPIPI + blk];
        P.scov[((size_t)s * TRI + tri(a, b)) * N + i] = v;
        M[a * LD + b] = v;
        M[b * LD + a] = v;
      }
    }
    ODEF_TEAM_FOR(k, D) {
      const double v = ms[k];
      nan_seen = nan_seen || !(v == v);
      P.smean[((size_t)s * D + k) * N + i] = v;
    }
    t.sync();
  }
  if (nan_seen) P.retcode[i] = 3;  // "NaNs after smoothing" (src/smoothing.jl:25)
}

}  // namespace odef
