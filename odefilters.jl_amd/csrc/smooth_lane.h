// RTS smoother, one lane per trajectory (small D: packed covariance <= ~80 doubles).
// src/smoothing.jl:4-63.  In preconditioned coordinates, with X = filter covariance of time i,
//   B = A X A' + sigma2 Q = L L'            (predict, src/smoothing.jl:38)
//   Z = B^-1 (S^s_+ - B) B^-1               (two in-place two-sided triangular transforms)
//   W = A' Z A                              (in-place block congruence)
//   m^s = m + X A' B^-1 (m^s_+ - A m)       (src/smoothing.jl:44, G = X A' B^-1 never formed)
//   S^s = X + X W X                         ( = X + G (S^s_+ - B) G', test/filtering.jl:113 )
// so no dense D x D matrix (G, I - G A) is ever live: the working set is three packed symmetric
// matrices, one of which (X, read-only) sits in lane-private LDS ([element][lane]: conflict-free,
// immediate-offset ds_read_b64).  78 doubles x 64 lanes x 8 B = 39 KB per wave at D = 12, so four
// waves (one per SIMD) fill the 160 KB of a CU exactly.
#pragma once
#include "ek_lane.h"

namespace odef {

// lane-private array in LDS (device: base = lds + lane, stride = 64) or plain memory (host: stride 1)
struct LaneMem {
  double* base;
  int stride;
  __device__ inline double get(int k) const { return base[k * stride]; }
  __device__ inline void set(int k, double v) const { base[k * stride] = v; }
};

// S <- L^-1 S L^-T  then  S <- L^-T S L^-1   (packed lower symmetric S, packed lower Cholesky factor L)
template <int D>
__device__ inline void two_sided_inverse(double (&S)[D * (D + 1) / 2], const double (&L)[D * (D + 1) / 2], const double (&dinv)[D]) {
  // forward: elementary congruences with L_k^-1
  static_for<0, D>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double inv = dinv[k];  // 1 / L_kk from the factorisation (0 for a zeroed column)
    const double tkk = S[tri(k, k)] * inv * inv;
    S[tri(k, k)] = tkk;
#pragma unroll
    for (int j = 0; j < k; ++j) S[tri(k, j)] *= inv;
    double b[D];
#pragma unroll
    for (int i = k + 1; i < D; ++i) {
      const double a = S[tri(i, k)] * inv;
      const double lik = L[tri(i, k)];
      b[i] = a - 0.5 * lik * tkk;
#pragma unroll
      for (int j = 0; j < k; ++j) S[tri(i, j)] -= lik * S[tri(k, j)];
    }
#pragma unroll
    for (int i = k + 1; i < D; ++i) {
      const double lik = L[tri(i, k)];
#pragma unroll
      for (int j = k + 1; j <= i; ++j) S[tri(i, j)] -= lik * b[j] + L[tri(j, k)] * b[i];
      S[tri(i, k)] = b[i] - 0.5 * lik * tkk;
    }
    ODEF_SCHED_FENCE();
  });
  // backward: elementary congruences with L_k^-T (only row/column k changes)
  static_for<0, D>([&](auto kc) {
    constexpr int k = D - 1 - decltype(kc)::value;
    const double inv = dinv[k];  // 1 / L_kk from the factorisation (0 for a zeroed column)
    double t[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double s = S[symidx(k, j)];
#pragma unroll
      for (int i = k + 1; i < D; ++i) s -= L[tri(i, k)] * S[symidx(i, j)];
      t[j] = s;
    }
    double tk = t[k];
#pragma unroll
    for (int i = k + 1; i < D; ++i) tk -= L[tri(i, k)] * t[i];
#pragma unroll
    for (int j = 0; j < D; ++j)
      if (j != k) S[symidx(k, j)] = t[j] * inv;
    S[tri(k, k)] = tk * inv * inv;
    ODEF_SCHED_FENCE();
  });
}

// X <- A' X A in place on packed symmetric storage (A = At (x) I_d): A' = E'_1 ... E'_{NB-1} with
// E'_J = I + sum_{j<J} At[j][J] e_J e_j' (block row J picks up the still-untouched block rows j < J).
template <int d, int NB>
__device__ inline void congruence_At_inplace(const PriorConsts& pc, double (&X)[d * NB * (d * NB + 1) / 2]) {
  constexpr int D = d * NB;
#pragma unroll
  for (int J = NB - 1; J >= 1; --J) {
    double Wd[d][d];
#pragma unroll
    for (int a = 0; a < d; ++a)
#pragma unroll
      for (int b = 0; b < d; ++b) {
        double t = X[symidx(J * d + a, J * d + b)];
#pragma unroll
        for (int j = 0; j < J; ++j) t += pc.At[j][J] * X[symidx(j * d + a, J * d + b)];
        Wd[a][b] = t;
      }
#pragma unroll
    for (int a = 0; a < d; ++a)
#pragma unroll
      for (int c = 0; c < D; ++c) {
        if (c / d == J) continue;
        double t = X[symidx(J * d + a, c)];
#pragma unroll
        for (int j = 0; j < J; ++j) t += pc.At[j][J] * X[symidx(j * d + a, c)];
        X[symidx(J * d + a, c)] = t;
      }
#pragma unroll
    for (int a = 0; a < d; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        double t = Wd[a][b];
#pragma unroll
        for (int j = 0; j < J; ++j) t += pc.At[j][J] * X[symidx(J * d + a, j * d + b)];
        X[tri(J * d + a, J * d + b)] = t;
      }
  }
}

// Core of one RTS step for one trajectory in preconditioned coordinates (src/smoothing.jl:31-63).
//   in : mt = P m (filter mean), B = P S P (filter covariance, packed; destroyed), xl holds the same P S P,
//        msn = P m^s_+, Cs = P S^s_+ P (packed; destroyed), sigma2, pij = diag of P^-1 per derivative block
//   out: ms_out (un-preconditioned smoothed mean), sink(k, v) for every packed entry k of the
//        un-preconditioned smoothed covariance, in storage order
template <int d, int NB, class CovSink>
__device__ inline void rts_step_core(const PriorConsts& pc, const double (&pij)[NB], const double (&mt)[d * NB],
                                     double (&B)[d * NB * (d * NB + 1) / 2], double (&Cs)[d * NB * (d * NB + 1) / 2],
                                     const double (&msn)[d * NB], double sigma2, const LaneMem& xl,
                                     double (&ms_out)[d * NB], CovSink& sink) {
  constexpr int D = d * NB, TRI = D * (D + 1) / 2;
  // predict (src/smoothing.jl:38)
  double dl[D];
#pragma unroll
  for (int J = 0; J < NB; ++J)
#pragma unroll
    for (int a = 0; a < d; ++a) {
      double t = mt[J * d + a];
#pragma unroll
      for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * mt[j * d + a];
      dl[J * d + a] = msn[J * d + a] - t;  // m^s_+ - m^-
    }
  ODEF_SCHED_FENCE();
  predict_cov_inplace<d, NB>(pc, B, sigma2);
  ODEF_SCHED_FENCE();
#pragma unroll
  for (int k = 0; k < TRI; ++k) Cs[k] -= B[k];  // M = S^s_+ - S^-
  int fixes = 0;
  double dinv[D];  // 1 / L_kk (0 for a zeroed column): every later division by a pivot is a multiplication with these
  chol_packed<D>(B, fixes, dinv);
  ODEF_SCHED_FENCE();
  // w = A' B^-1 delta ;  m^s = m + X w  (src/smoothing.jl:42-44)
#pragma unroll
  for (int k = 0; k < D; ++k) {
    double t = dl[k];
#pragma unroll
    for (int c = 0; c < k; ++c) t -= B[tri(k, c)] * dl[c];
    dl[k] = t * dinv[k];
  }
#pragma unroll
  for (int k = D - 1; k >= 0; --k) {
    double t = dl[k];
#pragma unroll
    for (int c = k + 1; c < D; ++c) t -= B[tri(c, k)] * dl[c];
    dl[k] = t * dinv[k];
  }
  double wv[D];
#pragma unroll
  for (int K = 0; K < NB; ++K)
#pragma unroll
    for (int b = 0; b < d; ++b) {
      double t = dl[K * d + b];
#pragma unroll
      for (int j = 0; j < K; ++j) t += pc.At[j][K] * dl[j * d + b];
      wv[K * d + b] = t;
    }
  // Z = B^-1 M B^-1 ;  W = A' Z A
  ODEF_SCHED_FENCE();
  two_sided_inverse<D>(Cs, B, dinv);
  ODEF_SCHED_FENCE();
  congruence_At_inplace<d, NB>(pc, Cs);
  ODEF_SCHED_FENCE();
  // rows of the result, handed out as they are produced
  static_for<0, D>([&](auto ac) {
    constexpr int a = decltype(ac)::value;
    double xa[D], u[D];
#pragma unroll
    for (int c = 0; c < D; ++c) xa[c] = xl.get(symidx(a, c));
    double mnew = mt[a];
#pragma unroll
    for (int c = 0; c < D; ++c) mnew += xa[c] * wv[c];
    ms_out[a] = mnew * pij[a / d];  // un-precondition (src/smoothing.jl:26)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) t += xa[k] * Cs[symidx(k, c)];
      u[c] = t;
    }
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      double t = xa[b];
#pragma unroll
      for (int k = 0; k < D; ++k) t += u[k] * xl.get(symidx(k, b));
      sink(tri(a, b), t * (pij[a / d] * pij[b / d]));
    }
    ODEF_SCHED_FENCE();
  });
}

// Largest value of v over the lanes of the wavefront that are inside the batch, as a wave-uniform scalar.
// Adaptive solves store a different number of records per trajectory; walking the SAME save slot in all lanes of
// a wavefront (lanes join when the slot reaches their own last record) keeps every record access a contiguous
// 512-byte row.  Walking each lane from its own end instead touched ~10 different slots per load instruction and
// ran the adaptive smoother 8x slower per step.
__device__ inline long wave_uniform_max(long v, bool valid) {
#ifdef ODEF_HOST_EMUL
  (void)valid;
  return v;
#else
  int x = valid ? (int)v : 0;
  for (int m = 32; m >= 1; m >>= 1) {
    const int y = __shfl_xor(x, m, 64);
    x = x > y ? x : y;
  }
  return (long)__builtin_amdgcn_readfirstlane(x);
#endif
}

// `n_hi`: wave-uniform upper bound of the record count of the lanes of this wavefront (fixed grid: n_save).
// ADAPT: per-trajectory record counts and step sizes (adaptive solve); otherwise everything about the grid is
// wave-uniform and comes from the host tables.
template <int d, int q, bool ADAPT>
__device__ inline void smooth_lane_v2(const SmoothParams& P, long i0, unsigned lane, const LaneMem& xl, long n_hi) {
  constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  const long i = i0 + lane;
  const long n = ADAPT ? (long)P.nsaved[i] : P.n_save;
  const size_t N = (size_t)P.N;
  const PriorConsts& pc = P.pc;
  // The smoothed moments of time s+1 are CARRIED in registers (un-preconditioned, exactly the values stored): every
  // record is read once and written once.  (Until round 2 the covariance was re-read from the record written one
  // iteration earlier, 1.5 x the algorithmic traffic: profiles/r01_smoother_lane_pmc.json.)
  double ms[D], Sn[TRI];
  for (int w = 0; w < 2; ++w) {  // first and last record are copied (src/smoothing.jl:11)
    const long s = w == 0 ? 0 : n - 1;
    {
      const double* src = P.cov + ((size_t)s * TRI) * N + i;
#pragma unroll
      for (int k = 0; k < TRI; ++k) Sn[k] = src[(size_t)k * N];  // all loads in flight before the first store
#pragma unroll
      for (int k = 0; k < D; ++k) ms[k] = P.mean[((size_t)s * D + k) * N + i];
      ODEF_SCHED_FENCE();
      double* dst = P.scov + ((size_t)s * TRI) * N + i;
#pragma unroll
      for (int k = 0; k < TRI; ++k) dst[(size_t)k * N] = Sn[k];
#pragma unroll
      for (int k = 0; k < D; ++k) P.smean[((size_t)s * D + k) * N + i] = ms[k];
    }
  }
  bool nan_seen = false;
  // The slot index walks the same value in every lane but is deliberately kept in a VGPR: with a scalar slot
  // the compiler moves the record address arithmetic to the SALU (s_mul chains, SGPR spills, hazard nops in
  // front of the loads) and the fixed-grid smoother measured 44 ms instead of 39 ms.  (Round 2 tried the other
  // extreme as well -- wave-uniform row bases advanced by s_add_u32 / s_addc_u32 and raw buffer accesses with a
  // constant lane offset: 616 fewer VALU address instructions, but as many scalar ones, and a lone wavefront per SIMD
  // pays one issue slot for either kind: 36 -> 60 ms with the spills that came with it.)
  long s_start = n_hi - 2;
#ifndef ODEF_HOST_EMUL
  {
    int lo = (int)s_start;
    asm volatile("" : "+v"(lo));
    s_start = lo;
  }
#endif
  for (long s = s_start; s >= 1; --s) {
    if constexpr (ADAPT) {
      if (s > n - 2) continue;  // this trajectory has fewer records: it joins at its own last one (ms, Sn hold it)
    }
    double h, pj[NB], pij[NB];
    if constexpr (ADAPT) {
      h = P.tsave[(size_t)(s + 1) * N + i] - P.tsave[(size_t)s * N + i];
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
    if (h == 0.0) {  // src/smoothing.jl:13-16
#pragma unroll
      for (int k = 0; k < D; ++k) P.smean[((size_t)s * D + k) * N + i] = ms[k];
#pragma unroll
      for (int k = 0; k < TRI; ++k) P.scov[((size_t)s * TRI + k) * N + i] = Sn[k];
      continue;
    }
    const double sigma2 = P.diff[(size_t)(s + 1) * N + i];
    // x_i and x_{i+1}^s in preconditioned coordinates (src/smoothing.jl:23-24)
    // Phase 1: every load of the step in flight (raw values, no arithmetic in between: a spill reload between two
    // loads would make the compiler wait for the first one -- scratch and global loads share vmcnt -- and serialise
    // the memory latencies of the step).  Phase 2: scale.
    double mt[D], B[TRI], Cs[TRI];
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
        const double pp = pj[a / d] * pj[b / d];
        const double x = B[tri(a, b)] * pp;
        xl.set(tri(a, b), x);
        B[tri(a, b)] = x;
        Cs[tri(a, b)] = Sn[tri(a, b)] * pp;
      }
    double msn[D], msnew[D];
#pragma unroll
    for (int k = 0; k < D; ++k) msn[k] = pj[k / d] * ms[k];
    auto sink = [&](int k, double v) {
      Sn[k] = v;
      P.scov[((size_t)s * TRI + k) * N + i] = v;
    };
    rts_step_core<d, NB>(pc, pij, mt, B, Cs, msn, sigma2, xl, msnew, sink);
#pragma unroll
    for (int k = 0; k < D; ++k) ms[k] = msnew[k];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      nan_seen = nan_seen || !(ms[k] == ms[k]);
      P.smean[((size_t)s * D + k) * N + i] = ms[k];
    }
  }
  if (nan_seen) P.retcode[i] = 3;  // "NaNs after smoothing" (src/smoothing.jl:25)
}

}  // namespace odef
