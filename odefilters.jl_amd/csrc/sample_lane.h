// Posterior sampling on the saved grid (src/solution_sampling.jl:24-62) and on a dense grid (:63-75).
// One lane per (sample, trajectory): x_N ~ N(mu_N, S_N); backwards x_i ~ smooth(x_filt[i], delta(x_{i+1}))
// in preconditioned coordinates, i.e. the RTS step core with a zero "next" covariance, followed by
// mean + L xi with L the lower-triangular factor of the un-preconditioned conditional covariance (the
// reference multiplies the noise with the R' of its QR, src/solution_sampling.jl:10 -- a different square
// root of the same covariance, so the same distribution).  The noise is the build's own counter-based
// stream (oracle/odefilter_oracle.py sample_normal): splitmix64 -> Box-Muller.
#pragma once
#include "smooth_lane.h"
#include "dense_lane.h"

namespace odef {

struct SampleParams {
  PriorConsts pc;
  long N;
  long n_save;          // fixed: number of saves; adaptive: capacity
  int adaptive;
  const double* ptab;   // fixed: preconditioner tables
  const int* tab_idx;   // fixed: [n_save-1]
  const double* hs;     // fixed: [n_save-1]
  const double* tsave;  // adaptive: [n_save][N]
  const int* nsaved;    // [N]
  const double* mean;   // filter records
  const double* cov;
  const double* diff;
  long n_samples;
  unsigned long long seed;
  double noise_scale;   // 1: samples; 0: the chain of conditional means (test hook)
  // dense-grid mode (dense_sample_states, src/solution_sampling.jl:63-69): tq != nullptr.  mean/cov are then the
  // FILTER posterior interpolated at the n_save shared times tq (dense_output_kernel), diff stays the solver's
  // record array, and the diffusion of an interval is looked up by time (:41) in the record times.
  const double* tq;     // [n_save]
  const double* rec_t;  // fixed solves: [n_rec] record times (adaptive solves: tsave / nsaved)
  long n_rec;
  double* samples;      // [n_save][D][n_samples][N]
};

__device__ inline unsigned long long splitmix64_dev(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// N(0,1) variate number c of the stream `seed`
__device__ inline double sample_normal(unsigned long long seed, unsigned long long c) {
  const double u1 = (double)(splitmix64_dev(seed + 2ull * c) >> 11) * 0x1.0p-53;
  const double u2 = (double)(splitmix64_dev(seed + 2ull * c + 1ull) >> 11) * 0x1.0p-53;
#ifdef ODEF_HOST_EMUL
  return sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2);
#else
  return sqrt(-2.0 * log(1.0 - u1)) * cospi(2.0 * u2);  // same value to rounding, without cos()'s large-argument reduction code
#endif
}

// out = m + scale * L xi, L = lower factor of the packed covariance C (destroyed)
template <int D>
__device__ inline void draw_packed(const double (&m)[D], double (&C)[D * (D + 1) / 2], double scale,
                                   unsigned long long seed, unsigned long long c0, double (&out)[D]) {
  int fixes = 0;
  chol_packed<D>(C, fixes);
  double xi[D];
#pragma unroll
  for (int k = 0; k < D; ++k) xi[k] = sample_normal(seed, c0 + (unsigned long long)k);
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int b = 0; b <= a; ++b) t += C[tri(a, b)] * xi[b];
    out[a] = m[a] + scale * t;
  }
}

template <int d, int q>
__device__ inline void sample_lane(const SampleParams& P, long i, long j, const LaneMem& xl, long n_hi) {
  constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  const size_t N = (size_t)P.N, NS = (size_t)P.n_samples;
  const bool dense = P.tq != nullptr;
  const long n = (P.adaptive && !dense) ? (long)P.nsaved[i] : P.n_save;
  const PriorConsts& pc = P.pc;
  auto out = [&](long s, int k) -> double& { return P.samples[(((size_t)s * D + k) * NS + (size_t)j) * N + i]; };
  auto ctr = [&](long s) { return (((unsigned long long)i * NS + (unsigned long long)j) * (unsigned long long)P.n_save + (unsigned long long)s) * (unsigned long long)D; };
  double xs[D];  // the sample of the later time
  {
    double m[D], C[TRI];
#pragma unroll
    for (int k = 0; k < D; ++k) m[k] = P.mean[((size_t)(n - 1) * D + k) * N + i];
#pragma unroll
    for (int k = 0; k < TRI; ++k) C[k] = P.cov[((size_t)(n - 1) * TRI + k) * N + i];
    draw_packed<D>(m, C, P.noise_scale, P.seed, ctr(n - 1), xs);
#pragma unroll
    for (int k = 0; k < D; ++k) out(n - 1, k) = xs[k];
  }
  for (long s = n_hi - 2; s >= 0; --s) {  // wave-uniform slot, see wave_uniform_max (smooth_lane.h)
    if (s > n - 2) continue;
    double h, pj[NB], pij[NB];
    if (P.adaptive || dense) {
      h = dense ? P.tq[s + 1] - P.tq[s] : P.tsave[(size_t)(s + 1) * N + i] - P.tsave[(size_t)s * N + i];
      double val = (h != 0.0) ? precond_val<q>(h) : 1.0;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        pj[J] = val;
        pij[J] = 1.0 / val;
        val *= h;
      }
    } else {
      h = P.hs[s];
      const double* __restrict__ tab = P.ptab + (size_t)P.tab_idx[s] * kTabStride;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        pj[J] = tab[kTabPJ + J];
        pij[J] = tab[kTabPIJ + J];
      }
    }
    if (h == 0.0) {  // duplicated save time: the state is the later one
#pragma unroll
      for (int k = 0; k < D; ++k) out(s, k) = xs[k];
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
    // all loads of the step first, arithmetic afterwards (see smooth_lane_v2)
    double mt[D], B[TRI], Cs[TRI], msn[D], mc[D];
    {
      const double* pc_ = P.cov + ((size_t)s * TRI) * N + i;
      const double* pm_ = P.mean + ((size_t)s * D) * N + i;
#pragma unroll
      for (int k = 0; k < TRI; ++k) {
        B[k] = *pc_;
        pc_ += N;
      }
#pragma unroll
      for (int k = 0; k < D; ++k) {
        mt[k] = *pm_;
        pm_ += N;
      }
    }
    ODEF_SCHED_FENCE();
#pragma unroll
    for (int k = 0; k < D; ++k) mt[k] *= pj[k / d];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        const double x = B[tri(a, b)] * (pj[a / d] * pj[b / d]);
        xl.set(tri(a, b), x);
        B[tri(a, b)] = x;
        Cs[tri(a, b)] = 0.0;  // Gaussian(sample, 0) (src/solution_sampling.jl:52)
      }
#pragma unroll
    for (int k = 0; k < D; ++k) msn[k] = pj[k / d] * xs[k];
    // B is dead inside the core once the covariance rows are being produced: it receives them
    auto sink = [&](int k, double v) { B[k] = v; };
    rts_step_core<d, NB>(pc, pij, mt, B, Cs, msn, sigma2, xl, mc, sink);
    draw_packed<D>(mc, B, P.noise_scale, P.seed, ctr(s), xs);
#pragma unroll
    for (int k = 0; k < D; ++k) out(s, k) = xs[k];
  }
}

}  // namespace odef
