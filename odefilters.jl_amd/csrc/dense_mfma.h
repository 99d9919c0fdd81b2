// Dense output / saveat for the workgroup-per-trajectory path (Pleiades, D = 28 (q+1) up to 168): the posterior at
// arbitrary times (src/solution.jl:165-210) -- locate the interval, `predict` from the left filter state with P(h1)
// and, for the smoothed posterior inside the grid, one `smooth` against the right smoothed state with P(h2) -- on the
// dense algebra of the MFMA smoother (smooth_mfma.h: mfma_predict_phase / mfma_gain_phase).  One workgroup per
// (trajectory, query time) item, walking the items with a grid stride over a bounded set of workspaces.
// Same semantics as the one-lane-per-item kernel for D <= 12 (dense_lane.h).
#pragma once
#include "dense_lane.h"
#include "smooth_mfma.h"

namespace odef {

template <int d, int q>
__device__ __attribute__((always_inline)) inline void dense_mfma_item(const DenseParams& P, long i, long jq, double* __restrict__ ws, double* __restrict__ lds) {
  constexpr int NB = q + 1;
  using W = MfmaSmoothWs<d, NB>;
  constexpr int D = W::D, DP = W::DP, TRI = D * (D + 1) / 2;
  const int tid = (int)threadIdx.x, nth = (int)blockDim.x;
  const size_t N = (size_t)P.N;
  const long n = P.adaptive ? (long)P.nsaved[i] : P.n_save;
  const double tval = P.tq[jq];
  double* X = ws + W::X;
  double* BM = ws + W::BM;
  double* SG = ws + W::SG;
  double* mf_ = lds + W::MF;
  double* ms_ = lds + W::MS;
  double* mp_ = lds + W::MP;
  double* pj_ = lds + W::PJ;
  double* pij_ = lds + W::PIJ;
  auto tat = [&](long s) { return P.adaptive ? P.tsave[(size_t)s * N + i] : P.tgrid[s]; };
  // idx = number of grid points <= tval (Julia's 1-based `sum(t .<= tval)`), by bisection: every thread the same walk
  long lo = 0, hi = n;
  while (lo < hi) {
    const long mid = (lo + hi) / 2;
    if (tat(mid) <= tval) lo = mid + 1;
    else hi = mid;
  }
  const long idx = lo, il = idx - 1;
  double* qm = P.qmean + ((size_t)jq * D) * N + i;
  double* qc = P.qcov + ((size_t)jq * TRI) * N + i;
  if (idx <= 0) {  // tval < t0: the reference throws "Invalid t<t0" (src/solution.jl:169-171)
    for (int k = tid; k < D; k += nth) qm[(size_t)k * N] = __builtin_nan("");
    for (int e = tid; e < TRI; e += nth) qc[(size_t)e * N] = __builtin_nan("");
    return;
  }
  const bool sm = P.smoothed != 0;
  if (tat(il) == tval) {  // src/solution.jl:172-176: exactly a stored time
    const double* m = (sm ? P.smean : P.mean) + ((size_t)il * D) * N + i;
    const double* c = (sm ? P.scov : P.cov) + ((size_t)il * TRI) * N + i;
    for (int k = tid; k < D; k += nth) qm[(size_t)k * N] = m[(size_t)k * N];
    for (int e = tid; e < TRI; e += nth) qc[(size_t)e * N] = c[(size_t)e * N];
    return;
  }
  // diffusions[min(idx, end)] (src/solution.jl:181): slot s holds the diffusion of step s-1 -> s
  const long sd = (idx < n - 1) ? idx : n - 1;
  const double sigma2 = P.diff[(size_t)sd * N + i];
  // extrapolate: goal_pred = P1^-1 predict(P1 prev, A, Qh)  (src/solution.jl:184-189)
  const double h1 = tval - tat(il);
  __syncthreads();  // the previous item of this workgroup is done with the LDS vectors
  if (tid < DP) {
    double pj = 0.0, pij = 0.0;
    if (tid < D) {
      double a[NB], b[NB];
      precond_from_h<q>(h1, a, b);
      pj = a[tid / d];
      pij = b[tid / d];
    }
    pj_[tid] = pj;
    pij_[tid] = pij;
  }
  __syncthreads();
  {
    TriWalk tw(tid);
    const double* src = P.cov + ((size_t)il * TRI) * N + i;
    for (int e = tid; e < TRI; e += nth, tw.advance(nth)) {
      const double v = src[(size_t)e * N] * (pj_[tw.a] * pj_[tw.b]);
      X[tw.a * DP + tw.b] = v;
      X[tw.b * DP + tw.a] = v;
    }
  }
  for (int k = tid; k < D; k += nth) {
    mf_[k] = pj_[k] * P.mean[((size_t)il * D + k) * N + i];
    ms_[k] = 0.0;
  }
  __syncthreads();
  mfma_predict_phase<d, q>(P.pc, sigma2, ws, lds);  // BM = A X A' + sigma2 Q, mp_ = A mf_ (preconditioned with P1)
  if (!sm || il >= n - 1) {  // filter posterior, or beyond the last time (src/solution.jl:191-193)
    for (int k = tid; k < D; k += nth) qm[(size_t)k * N] = pij_[k] * mp_[k];
    TriWalk tw(tid);
    for (int e = tid; e < TRI; e += nth, tw.advance(nth)) qc[(size_t)e * N] = BM[tw.a * DP + tw.b] * (pij_[tw.a] * pij_[tw.b]);
    return;
  }
  // smooth against x_smooth[idx+1] with P(h2)  (src/solution.jl:199-209): the predicted state plays the filter state
  const double h2 = tat(il + 1) - tval;
  double f = 0.0, pj2 = 0.0, pij2 = 0.0;  // this thread's component: P2 P1^-1, P2, P2^-1
  if (tid < D) {
    double a[NB], b[NB];
    precond_from_h<q>(h2, a, b);
    pj2 = a[tid / d];
    pij2 = b[tid / d];
    f = pj2 * pij_[tid];
    mf_[tid] = pj2 * (pij_[tid] * mp_[tid]);
    ms_[tid] = P.smean[((size_t)(il + 1) * D + tid) * N + i];
  }
  __syncthreads();  // every thread has read pij_ (P1^-1) for its component
  if (tid < DP) {
    pj_[tid] = pj2;    // from here on the vectors hold P2 ...
    pij_[tid] = pij2;
    mp_[tid] = f;      // ... and mp_ (free until the next predict phase) the factors P2 P1^-1
  }
  __syncthreads();
  for (int e = tid; e < D * D; e += nth) {
    const int r = e / D, c = e % D;
    X[r * DP + c] = BM[r * DP + c] * (mp_[r] * mp_[c]);
  }
  {
    TriWalk tw(tid);
    const double* src = P.scov + ((size_t)(il + 1) * TRI) * N + i;
    for (int e = tid; e < TRI; e += nth, tw.advance(nth)) {
      const double v = src[(size_t)e * N];
      SG[tw.a * DP + tw.b] = v;
      SG[tw.b * DP + tw.a] = v;
    }
  }
  __syncthreads();
  mfma_predict_phase<d, q>(P.pc, sigma2, ws, lds);
  mfma_gain_phase<d, q>(ws, lds);
  for (int k = tid; k < D; k += nth) qm[(size_t)k * N] = ms_[k];
  {
    TriWalk tw(tid);
    for (int e = tid; e < TRI; e += nth, tw.advance(nth))
      qc[(size_t)e * N] = (X[tw.a * DP + tw.b] + BM[tw.a * DP + tw.b]) * (pij_[tw.a] * pij_[tw.b]);
  }
}

}  // namespace odef
