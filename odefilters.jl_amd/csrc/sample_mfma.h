// Posterior sampling for the workgroup-per-trajectory path (Pleiades, D = 28 (q+1) up to 168; src/solution_sampling.jl:24-75) on
// the dense algebra of the MFMA smoother (smooth_mfma.h): one workgroup per (trajectory, sample) item.  Same scheme, counters and
// noise stream as sample_lane.h / sample_rows.h: x_N ~ N(mu_N, S_N); backwards x_i ~ smooth(x_filt[i], delta(x_{i+1})) --
// mfma_predict_phase / mfma_gain_phase with a zero "next" covariance -- and mean + L xi with L the lower-triangular factor of the
// conditional covariance from the blocked Cholesky of mfma_dense.h (non-positive pivots zero their column).
#pragma once
#include "sample_lane.h"
#include "smooth_mfma.h"

namespace odef {

// ms_ <- ms_ + scale L xi for the covariance C (full symmetric, un-preconditioned) in the X slot of the workspace; variates
// c0 .. c0 + D - 1 of the stream.  X is destroyed (its padding block ends as the identity, which nothing reads).
template <int d, int q>
__device__ __attribute__((always_inline)) inline void mfma_draw(double scale, unsigned long long seed, unsigned long long c0, double* __restrict__ ws, double* __restrict__ lds) {
  using W = MfmaSmoothWs<d, q + 1>;
  constexpr int D = W::D, DP = W::DP, DPB = W::DPB;
  const int tid = (int)threadIdx.x, nth = (int)blockDim.x;
  double* X = ws + W::X;
  double* LM = ws + W::LM;
  double* ms_ = lds + W::MS;
  double* dl_ = lds + W::DL;
  for (int e = tid; e < (DP - D) * DP; e += nth) {  // identity in the padding block: the factorisation runs over DP x DP
    const int r = D + e / DP, c = e % DP;
    X[r * DP + c] = (r == c) ? 1.0 : 0.0;
    X[c * DP + r] = (r == c) ? 1.0 : 0.0;
  }
  for (int k = tid; k < D; k += nth) dl_[k] = sample_normal(seed, c0 + (unsigned long long)k);
  __syncthreads();
  mf::wg_cholesky_upper<DPB>(X, LM, DP, lds);  // LM = L (lower, row-major)
  __syncthreads();
  for (int a = tid; a < D; a += nth) {
    double acc = 0.0;
#pragma unroll 4
    for (int b = 0; b <= a; ++b) acc += LM[a * DP + b] * dl_[b];
    ms_[a] += scale * acc;
  }
  __syncthreads();
}

template <int d, int q>
__device__ __attribute__((always_inline)) inline void sample_mfma_item(const SampleParams& P, long i, long j, double* __restrict__ ws, double* __restrict__ lds) {
  constexpr int NB = q + 1;
  using W = MfmaSmoothWs<d, NB>;
  constexpr int D = W::D, DP = W::DP, TRI = D * (D + 1) / 2;
  const int tid = (int)threadIdx.x, nth = (int)blockDim.x;
  const size_t N = (size_t)P.N, NS = (size_t)P.n_samples;
  const bool dense = P.tq != nullptr;
  const long n = (P.adaptive && !dense) ? (long)P.nsaved[i] : P.n_save;
  double* X = ws + W::X;
  double* BM = ws + W::BM;
  double* mf_ = lds + W::MF;
  double* ms_ = lds + W::MS;
  double* pj_ = lds + W::PJ;
  double* pij_ = lds + W::PIJ;
  auto out = [&](long s, int k) -> double& { return P.samples[(((size_t)s * D + k) * NS + (size_t)j) * N + i]; };
  auto ctr = [&](long s) { return (((unsigned long long)i * NS + (unsigned long long)j) * (unsigned long long)P.n_save + (unsigned long long)s) * (unsigned long long)D; };
  __syncthreads();  // the previous item of this workgroup is done with the LDS vectors
  // x_N ~ N(mu_N, S_N)
  {
    TriWalk tw(tid);
    const double* src = P.cov + ((size_t)(n - 1) * TRI) * N + i;
    for (int e = tid; e < TRI; e += nth, tw.advance(nth)) {
      const double v = src[(size_t)e * N];
      X[tw.a * DP + tw.b] = v;
      X[tw.b * DP + tw.a] = v;
    }
  }
  for (int k = tid; k < D; k += nth) ms_[k] = P.mean[((size_t)(n - 1) * D + k) * N + i];
  __syncthreads();
  mfma_draw<d, q>(P.noise_scale, P.seed, ctr(n - 1), ws, lds);
  for (int k = tid; k < D; k += nth) out(n - 1, k) = ms_[k];
  for (long s = n - 2; s >= 0; --s) {
    double h;
    if (P.adaptive || dense) h = dense ? P.tq[s + 1] - P.tq[s] : P.tsave[(size_t)(s + 1) * N + i] - P.tsave[(size_t)s * N + i];
    else h = uniform_load(P.hs + s);
    if (h == 0.0) {  // duplicated save time: the state is the later one
      for (int k = tid; k < D; k += nth) out(s, k) = ms_[k];
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
    __syncthreads();
    if (tid < DP) {  // preconditioner of this step, per state component
      double pj = 0.0, pij = 0.0;
      if (tid < D) {
        if (P.adaptive || dense) {
          double a[NB], b[NB];
          precond_from_h<q>(h, a, b);
          pj = a[tid / d];
          pij = b[tid / d];
        } else {
          const GlobalTab tab{P.ptab + (size_t)uniform_load(P.tab_idx + s) * kTabStride};
          pj = tab[kTabPJ + tid / d];
          pij = tab[kTabPIJ + tid / d];
        }
      }
      pj_[tid] = pj;
      pij_[tid] = pij;
    }
    __syncthreads();
    {  // the filter state of slot s, preconditioned; the later sample (ms_) is the "smoothed next state", its covariance (SG) zero
      TriWalk tw(tid);
      const double* src = P.cov + ((size_t)s * TRI) * N + i;
      for (int e = tid; e < TRI; e += nth, tw.advance(nth)) {
        const double v = src[(size_t)e * N] * (pj_[tw.a] * pj_[tw.b]);
        X[tw.a * DP + tw.b] = v;
        X[tw.b * DP + tw.a] = v;
      }
    }
    for (int k = tid; k < D; k += nth) mf_[k] = pj_[k] * P.mean[((size_t)s * D + k) * N + i];
    __syncthreads();
    mfma_predict_phase<d, q>(P.pc, sigma2, ws, lds);
    mfma_gain_phase<d, q>(ws, lds);  // ms_ = conditional mean (un-preconditioned), BM = G M G' = -G S^- G'
    __syncthreads();
    for (int e = tid; e < D * D; e += nth) {  // the conditional covariance, un-preconditioned, into the X slot
      const int r = e / D, c = e % D;
      X[r * DP + c] = (X[r * DP + c] + BM[r * DP + c]) * (pij_[r] * pij_[c]);
    }
    __syncthreads();
    mfma_draw<d, q>(P.noise_scale, P.seed, ctr(s), ws, lds);
    for (int k = tid; k < D; k += nth) out(s, k) = ms_[k];
  }
}

}  // namespace odef
