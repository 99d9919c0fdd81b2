// Posterior sampling for 12 < D <= 32 (src/solution_sampling.jl:24-75) on the row-per-lane teams of smooth_rows.h: one team of
// 16 / 32 lanes per (trajectory, sample).  Same scheme, counters and noise stream as sample_lane.h (D <= 12):
// x_N ~ N(mu_N, S_N); backwards x_i ~ smooth(x_filt[i], delta(x_{i+1})) -- rows_predict_phase / rows_gain_phase with a zero
// "next" covariance -- and mean + L xi with L = L_unit sqrt(D) the lower-triangular factor of the conditional covariance.
#pragma once
#include "sample_lane.h"
#include "smooth_rows.h"

namespace odef {

// x_r = m_r + scale (L xi)_r for the covariance whose row r sits in L.csr (un-preconditioned) and the mean in L.ms; the draw lands
// in L.ms.  Variates c0 .. c0 + D - 1 of the stream.
template <int d, int q, int TEAM>
__device__ inline void rows_draw(double scale, unsigned long long seed, unsigned long long c0, int tid, double* __restrict__ ws,
                                 RowState<d*(q + 1)>* st) {
  constexpr int NB = q + 1, D = d * NB;
  using W = RowsWs<d, NB>;
  const Team<TEAM> t{tid};
  (void)t;
  double* COL = ws + W::COL;
  double* DINV = ws + W::DINV;
  double* VMT = ws + W::VMT;
  ODEF_ROWS_PHASE(
    if (r < D) {
_Pragma("unroll")
      for (int c = 0; c < D; ++c) L.lr[c] = L.csr[c];
    }
  )
  rows_ldl<D, TEAM>(tid, COL, DINV, st);
  ODEF_ROWS_PHASE(
    if (r < D) {
      const double piv = L.lr[r];
      VMT[r] = (piv > 0.0) ? sqrt(piv) * sample_normal(seed, c0 + (unsigned long long)r) : 0.0;  // zero-pivot rule: no noise in that direction
    }
  )
  ODEF_ROWS_PHASE(
    if (r < D) {
      double acc = VMT[r];
_Pragma("unroll")
      for (int c = 0; c < D; ++c)
        if (c < r) acc += L.lr[c] * VMT[c];
      L.ms = L.ms + scale * acc;
    }
  )
}

template <int d, int q, int TEAM>
__device__ inline void sample_rows_lane(const SampleParams& P, long i, long j, int tid, double* __restrict__ ws, RowState<d*(q + 1)>* st) {
  constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  const size_t N = (size_t)P.N, NS = (size_t)P.n_samples;
  const bool dense = P.tq != nullptr;
  const long n = (P.adaptive && !dense) ? (long)P.nsaved[i] : P.n_save;
  const PriorConsts& pc = P.pc;
  const Team<TEAM> t{tid};
  (void)t;
  auto out = [&](long s, int k) -> double& { return P.samples[(((size_t)s * D + k) * NS + (size_t)j) * N + i]; };
  auto ctr = [&](long s) { return (((unsigned long long)i * NS + (unsigned long long)j) * (unsigned long long)P.n_save + (unsigned long long)s) * (unsigned long long)D; };
  // lane constants; x_N ~ N(mu_N, S_N)
  ODEF_ROWS_PHASE(
    if (r < D) {
_Pragma("unroll")
      for (int J = 0; J < NB; ++J) {
        if (J == r / d) {
_Pragma("unroll")
          for (int jj = 0; jj < NB; ++jj) {
            L.atr[jj] = pc.At[J][jj];
            L.qtr[jj] = pc.Qt[J][jj];
          }
        }
      }
      L.ms = P.mean[((size_t)(n - 1) * D + r) * N + i];
_Pragma("unroll")
      for (int c = 0; c < D; ++c) L.csr[c] = P.cov[((size_t)(n - 1) * TRI + symidx(r, c)) * N + i];
    }
  )
  rows_draw<d, q, TEAM>(P.noise_scale, P.seed, ctr(n - 1), tid, ws, st);
  ODEF_ROWS_PHASE(
    if (r < D) out(n - 1, r) = L.ms;
  )
  for (long s = n - 2; s >= 0; --s) {
    double h;
    double pjv[NB], pijv[NB];
    if (P.adaptive || dense) {
      h = dense ? P.tq[s + 1] - P.tq[s] : P.tsave[(size_t)(s + 1) * N + i] - P.tsave[(size_t)s * N + i];
      double val = (h != 0.0) ? precond_val<q>(h) : 1.0;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        pjv[J] = val;
        pijv[J] = 1.0 / val;
        val *= h;
      }
    } else {
      h = P.hs[s];
      const double* __restrict__ tab = P.ptab + (size_t)P.tab_idx[s] * kTabStride;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        pjv[J] = tab[kTabPJ + J];
        pijv[J] = tab[kTabPIJ + J];
      }
    }
    if (h == 0.0) {  // duplicated save time: the state is the later one
      ODEF_ROWS_PHASE(
        if (r < D) out(s, r) = L.ms;
      )
      continue;
    }
    long sd = s + 1;  // slot k holds the diffusion of the step k-1 -> k
    if (dense) {      // i_diffusion = sum(difftimes .<= ts[i]) (src/solution_sampling.jl:41), by bisection
      const long nrec = P.adaptive ? (long)P.nsaved[i] : P.n_rec;
      const double tval = P.tq[s];
      long lo = 0, hi = nrec;
      while (lo < hi) {
        const long mid = (lo + hi) / 2;
        const double tm = P.adaptive ? P.tsave[(size_t)mid * N + i] : P.rec_t[mid];
        if (tm <= tval) lo = mid + 1;
        else hi = mid;
      }
      sd = lo < nrec - 1 ? lo : nrec - 1;
      if (sd < 1) sd = nrec > 1 ? 1 : 0;
    }
    const double sigma2 = P.diff[(size_t)sd * N + i];
    // the filter state of slot s, preconditioned; the later sample (in L.ms) is the "smoothed next state" with zero covariance
    ODEF_ROWS_PHASE(
      if (r < D) {
        double pj_r = pjv[0];
        double pij_r = pijv[0];
_Pragma("unroll")
        for (int J = 1; J < NB; ++J) {
          pj_r = (r / d == J) ? pjv[J] : pj_r;
          pij_r = (r / d == J) ? pijv[J] : pij_r;
        }
        L.pj = pj_r;
        L.pij = pij_r;
        L.mf = pj_r * P.mean[((size_t)s * D + r) * N + i];
_Pragma("unroll")
        for (int c = 0; c < D; ++c) {
          L.xr[c] = P.cov[((size_t)s * TRI + symidx(r, c)) * N + i] * (pj_r * pjv[c / d]);
          L.csr[c] = 0.0;  // Gaussian(sample, 0) (src/solution_sampling.jl:52)
        }
      }
    )
    rows_predict_phase<d, q, TEAM>(pc, pjv, sigma2, tid, ws, st);
    rows_gain_phase<d, q, TEAM>(pijv, tid, ws, st);  // L.ms = conditional mean, L.csr = row of the conditional covariance
    rows_draw<d, q, TEAM>(P.noise_scale, P.seed, ctr(s), tid, ws, st);
    ODEF_ROWS_PHASE(
      if (r < D) out(s, r) = L.ms;
    )
  }
}

}  // namespace odef
