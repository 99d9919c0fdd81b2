// First kernel of a record of the D = 168 smoother's split pass (src/smoothing.jl:11-35): unpack and predict.
//
//   X = P Sigma_s P              the filter record of the stage (packed lower triangle), staged ONCE into LDS
//   Y' = A X,  B = A X A' + sigma^2 Q,  m^- = A P m,  delta = P m^s_{s+1} - m^-
//   (M = P Sigma^s_{s+1} P - B is formed by the on-chip kernel, from the Sigma^s tiles it wrote itself one record earlier)
//
// The prior is A = At (x) I_d (src/priors.jl): for every pair of components (a, b) the (q+1) x (q+1) block
// X_ab[j][k] = X[(j, a), (k, b)] maps onto the same block of Y' and B -- Y'_ab = At X_ab, B_ab = Y'_ab At' (+ sigma^2 Qt on
// a = b).  One thread per pair: 36 values out of LDS, two small triangular products in registers, and each result goes to the
// workspace once (runs of d contiguous doubles across the lanes of a wavefront).  Neither X nor Y' is read back: the
// workspace kernel this replaces for the split pass (mfma_predict_phase, smooth_mfma.h, still what the persistent kernel,
// dense output and sampling run) wrote X, re-read it for Y', re-read Y' for B -- 1.35 MB written and 1.15 MB fetched per
// trajectory and record (rocprofv3 PMC), HBM-bound at 0.75 ms per record of 2 048 trajectories.
//
// B and Y' leave tile-major (MfmaSmoothWs::tm), B as upper tiles only (what rts_smooth_sweeps_kernel loads), X not at all: the on-chip kernel reads the
// record itself for Sigma^s = X + G M G'.  Same arithmetic, term by term, as mfma_predict_phase.
#pragma once
#include "smooth_mfma.h"

namespace odef {

template <int d>
constexpr int predict_block() {
#ifdef ODEF_PREDICT_BLOCK  // (A/B builds of tools/split_smooth_stamps.hip)
  return ODEF_PREDICT_BLOCK;
#else
  // at most two wavefronts per SIMD: a pair's (q+1) x (q+1) block and its results want ~200 registers at q = 5 (with 13
  // wavefronts -- one round over the 784 pairs of d = 28, 128 registers -- 276 bytes per lane spill: 0.59 against 0.47 ms)
  return d * d >= 512 ? 512 : (d * d + 63) / 64 * 64;
#endif
}

template <int d, int q>
__device__ __attribute__((always_inline)) inline void smooth_predict_record(const SmoothParams& P, long i, double* __restrict__ ws, double* __restrict__ lds) {
  constexpr int NB = q + 1;
  using W = MfmaSmoothWs<d, NB>;
  constexpr int D = W::D, DP = W::DP, TRI = D * (D + 1) / 2;
  const int tid = (int)threadIdx.x, nth = (int)blockDim.x;
  const size_t N = (size_t)P.N;
  const long s = P.split_sa;
  const long n = P.adaptive ? (long)P.nsaved[i] : P.n_save;
  // the records of this trajectory that this block of the stage smooths (smooth_mfma_traj)
  const long s_hi = n - 2 < P.stage_hi ? n - 2 : P.stage_hi, s_lo = P.stage_s0 > 1 ? P.stage_s0 : 1;
  if (s < s_lo || s > s_hi) {  // (workgroup-uniform)
    if (tid == 0) ws[W::FLG] = -1.0;
    return;
  }
  double* rec = P.stage + ((size_t)(s - P.stage_s0) * N + (size_t)i) * (size_t)P.stage_ld;
  double* xs = lds;                 // the packed record
  double* mf_ = lds + TRI;          // P m
  double* ms_ = mf_ + DP;           // m^s_{s+1} (carried, un-preconditioned)
  double* SG = ws + W::SG;
  double h;
  if (P.adaptive) h = P.tsave[(size_t)(s + 1) * N + i] - P.tsave[(size_t)s * N + i];
  else h = uniform_load(P.hs + s);
  if (h == 0.0) {  // src/smoothing.jl:13-16: a repeated save time, the smoothed state carries over
    for (int k = tid; k < D; k += nth) P.smean[((size_t)s * D + k) * N + i] = ws[W::MSV + k];
    TriWalk tw(tid);
    for (int e = tid; e < TRI; e += nth, tw.advance(nth)) rec[e] = SG[W::tm(tw.b, tw.a)];  // (carried as upper tiles)
    if (tid == 0) ws[W::FLG] = -1.0;
    return;
  }
  // preconditioner of this step (src/preconditioning.jl:1-17), per derivative block
  double pjb[NB], pijb[NB];
  if (P.adaptive) {
    double tabv[kTabStride];
    precond_table_fast<q, NB>(h, tabv);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      pjb[j] = tabv[kTabPJ + j];
      pijb[j] = tabv[kTabPIJ + j];
    }
  } else {
    const GlobalTab tab{P.ptab + (size_t)uniform_load(P.tab_idx + s) * kTabStride};
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      pjb[j] = tab[kTabPJ + j];
      pijb[j] = tab[kTabPIJ + j];
    }
  }
  const double sigma2 = P.diff[(size_t)(s + 1) * N + i];
  const PriorConsts& pc = P.pc;
  for (int e = tid; e < TRI; e += nth) xs[e] = rec[e];
  for (int k = tid; k < D; k += nth) {
    double pj = 0.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) pj = (k / d == j) ? pjb[j] : pj;
    mf_[k] = pj * P.mean[((size_t)s * D + k) * N + i];
    ms_[k] = ws[W::MSV + k];
  }
  __syncthreads();
  // m^- = A P m (src/filtering.jl:22-25), delta; the vectors the on-chip kernel needs
  for (int k = tid; k < DP; k += nth) {
    double mfv = 0.0, dlv = 0.0, pj = 0.0, pij = 0.0;
    if (k < D) {
      const int J = k / d, a = k % d;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        pj = (J == j) ? pjb[j] : pj;
        pij = (J == j) ? pijb[j] : pij;
      }
      double t = mf_[k];
      for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * mf_[j * d + a];
      mfv = mf_[k];
      dlv = pj * ms_[k] - t;
    }
    ws[W::MFV + k] = mfv;
    ws[W::DLV + k] = dlv;
    ws[W::PIJV + k] = pij;
    ws[W::PJV + k] = pj;
  }
  // the padding block of B is the identity (nothing else writes there; Sigma^s and Y' are zero there from the set-up)
  double* YT = ws + W::YT;
  const bool yfromx = (d % 4 == 0) && P.split_sc == 1;  // the on-chip kernel forms Y' = A X itself (from the record, in registers)
  double* BM = ws + W::BM;
  for (int k = D + tid; k < DP; k += nth) BM[W::tm(k, k)] = 1.0;
  for (int it = tid; it < d * d; it += nth) {
    const int a = it / d, b = it % d;
    double x[NB][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int r = j * d + a, c = k * d + b;
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        x[j][k] = xs[hi * (hi + 1) / 2 + lo] * (pjb[j] * pjb[k]);
      }
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int r = J * d + a;
      double y[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        double t = x[J][k];
#pragma unroll
        for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * x[j][k];
        y[k] = t;
        if (!yfromx) YT[W::tm(r, k * d + b)] = t;
      }
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        double bv = y[K];
#pragma unroll
        for (int k = K + 1; k < NB; ++k) bv += pc.At[K][k] * y[k];
        if (a == b) bv += sigma2 * pc.Qt[J][K];
        const int c = K * d + b;
        if ((c >> 4) >= (r >> 4)) BM[W::tm(r, c)] = bv;  // (the tiles on and above the diagonal: all that is read)
      }
    }
  }
  if (tid == 0) ws[W::FLG] = (double)s;
}

template <int d, int q>
__global__ __launch_bounds__(predict_block<d>()) void rts_smooth_predict_kernel(const SmoothParams P, double* ws) {
  using W = MfmaSmoothWs<d, q + 1>;
  extern __shared__ double lds[];
  const long i = team_traj(P.N);
  if (i < 0) return;
  smooth_predict_record<d, q>(P, i, ws + (size_t)i * W::size, lds);
}

}  // namespace odef
