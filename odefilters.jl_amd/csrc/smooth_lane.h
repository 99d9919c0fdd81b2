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
    const double htkk = -0.5 * tkk;
#pragma unroll
    for (int j = 0; j < k; ++j) S[tri(k, j)] *= inv;
    double b[D];
#pragma unroll
    for (int i = k + 1; i < D; ++i) {
      const double lik = L[tri(i, k)];
      b[i] = __builtin_fma(lik, htkk, S[tri(i, k)] * inv);
#pragma unroll
      for (int j = 0; j < k; ++j) S[tri(i, j)] = __builtin_fma(-lik, S[tri(k, j)], S[tri(i, j)]);
    }
    // (two fused multiply-adds per entry: written as `S -= lik * b[j] + ljk * b[i]` the compiler must keep the rounding of
    // the inner sum, i.e. a multiply, an FMA and an add -- 286 instructions more per step at D = 12)
#pragma unroll
    for (int i = k + 1; i < D; ++i) {
      const double lik = L[tri(i, k)];
#pragma unroll
      for (int j = k + 1; j <= i; ++j)
        S[tri(i, j)] = __builtin_fma(-L[tri(j, k)], b[i], __builtin_fma(-lik, b[j], S[tri(i, j)]));
      S[tri(i, k)] = __builtin_fma(lik, htkk, b[i]);
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

// ---- explicit register discipline of the smoother loop ------------------------------------------------------------------
// A step needs three packed D x D matrices alive (78 doubles each at D = 12): the filter covariance X (read-only, in
// lane-private LDS), the factor L of the predicted covariance, and M -> Z -> W; the smoothed covariance carried to the
// next step takes over from L once that is dead.  Two matrices in registers are 312 of the 256 architectural VGPRs, so
// one of them has to sit in the AGPR half of the lane's 512 registers -- which VALU instructions cannot address: every
// visit costs a v_accvgpr_read/write per 32 bits.  Left to the register allocator this was 2 750 such moves per step, a
// quarter of the step's issue slots (profiles/r02_isa_counts.txt): it cannot know that the two-sided transforms rewrite
// all of S once per pivot but read only ONE COLUMN of L per pivot.  Handing the allocator AGPR-constrained values
// (inline assembly, "a" constraints) did not work either: it could not pack the ~700 short live ranges into 256
// registers and spilled AGPRs to scratch.  So the AGPR half is used as a hand-managed register FILE with static slots:
//   TRI slots               one packed matrix: carried S^s_+ (raw, across the prediction and the Cholesky factorisation)
//                           ->  L (across the two-sided transforms, read one column per pivot)  ->  the new S^s, entry
//                           by entry as it is produced
//   D slots                 carried m^s_+
// and everything the VALU works on stays below 256 VGPRs, so that the compiler never touches an AGPR itself --
// tests/test_build_hygiene.py checks exactly that on the ISA of this kernel (every AGPR reference inside these
// helpers' assembly).  850 moves per step instead of 2 750.  Host emulation: the file is a plain array.
#ifdef ODEF_HOST_EMUL
inline double* agpr_file() {
  static thread_local double f[256];
  return f;
}
template <int R>
inline void aput(double v) { agpr_file()[R] = v; }
template <int R>
inline double aget() { return agpr_file()[R]; }
inline void agpr_file_claim() {}
#else
template <int R>
__device__ __attribute__((always_inline)) inline void aput(double v) {
  static_assert(R >= 0 && R < 128, "a lane has 256 AGPRs = 128 doubles");
  asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(__double2loint(v)), "n"(2 * R));
  asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(__double2hiint(v)), "n"(2 * R + 1));
}
template <int R>
__device__ __attribute__((always_inline)) inline double aget() {
  static_assert(R >= 0 && R < 128, "a lane has 256 AGPRs = 128 doubles");
  int lo, hi;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(lo) : "n"(2 * R));
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(hi) : "n"(2 * R + 1));
  return __hiloint2double(hi, lo);
}
// Tells the compiler that the kernel uses the whole AGPR file (the kernel descriptor then allocates it).
__device__ __attribute__((always_inline)) inline void agpr_file_claim() { asm volatile("" ::: "a255"); }
#endif

// Pins the values of an array at this point of the program: an empty volatile assembly statement per element that
// "reads and writes" it.  __builtin_amdgcn_sched_barrier only stops the machine scheduler; instruction selection
// orders the PURE arithmetic of a basic block by itself and let the Cholesky factorisation start above the statements
// that still needed the unfactored matrix -- both generations of the matrix alive, the compiler reaching for AGPRs.
// Arithmetic on pinned values cannot be placed before the pin, and pins keep their order among themselves and with the
// AGPR-file accesses.
template <int n>
__device__ __attribute__((always_inline)) inline void pin(double (&a)[n]) {
#ifndef ODEF_HOST_EMUL
#pragma unroll
  for (int k = 0; k < n; ++k) asm volatile("" : "+v"(a[k]));
#else
  (void)a;
#endif
}

// In-place B = L D L' of a packed symmetric matrix: unit lower L below the diagonal (the diagonal entries are left as the pivots
// D_k), dinv[k] = 1 / D_k.  A non-positive pivot zeroes its column and gets the reciprocal 0 -- the rule of chol_packed
// (ek_math.h), i.e. the factor the reference's QR fallback yields for a positive semi-definite matrix (src/filtering.jl:38-47).
// No square roots: 12 pivots x (v_rcp_f64 + two Newton steps) instead of 12 x the 14-instruction sqrt / rsqrt sequence.
template <int D>
__device__ inline void ldl_packed_lane(double (&B)[D * (D + 1) / 2], double (&dinv)[D]) {
  static_for<0, D>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double piv = B[tri(k, k)];
    const bool ok = piv > 0.0;
    const double inv = ok ? rcp_pos(ok ? piv : 1.0) : 0.0;
    dinv[k] = inv;
    double v[D];
#pragma unroll
    for (int i = k + 1; i < D; ++i) {
      v[i] = B[tri(i, k)];
      B[tri(i, k)] = v[i] * inv;
    }
#pragma unroll
    for (int j = k + 1; j < D; ++j)
#pragma unroll
      for (int i = j; i < D; ++i) B[tri(i, j)] = __builtin_fma(-B[tri(i, k)], v[j], B[tri(i, j)]);
  });
}

// S <- B^-1 (S - B) B^-1 for B = L D L' with the factor in the AGPR file (slots LS + tri(i, k) hold the unit lower L, the DIAGONAL
// slots 1 / D_k): column k of L is fetched once per pivot and pass.
//   forward, unit lower:      S <- L^-1 S L^-T         (no scaling: the elementary congruences of a unit factor)
//   middle:                   S <- D^-1 S D^-1 - D^-1   = D^-1 (L^-1 (S - B) L^-T) D^-1, since L^-1 B L^-T = D: the difference
//                             S^s_+ - S^- the smoother needs is taken HERE, on the whitened matrix, so that it never has to be
//                             formed (and parked) before the factorisation overwrites S^-.  A zeroed column (1 / D_k taken as 0)
//                             zeroes its row and column and subtracts nothing.
//   backward, unit upper:     S <- L^-T S L^-1
template <int D, int LS>
__device__ inline void two_sided_inverse_parked(double (&S)[D * (D + 1) / 2]) {
  static_for<0, D>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    double lc[D];
    static_for<k + 1, D>([&](auto ic) { lc[decltype(ic)::value] = aget<LS + tri(decltype(ic)::value, k)>(); });
    const double htkk = -0.5 * S[tri(k, k)];
    double b[D];
#pragma unroll
    for (int i = k + 1; i < D; ++i) {
      b[i] = __builtin_fma(lc[i], htkk, S[tri(i, k)]);
#pragma unroll
      for (int j = 0; j < k; ++j) S[tri(i, j)] = __builtin_fma(-lc[i], S[tri(k, j)], S[tri(i, j)]);
    }
#pragma unroll
    for (int i = k + 1; i < D; ++i) {
#pragma unroll
      for (int j = k + 1; j <= i; ++j) S[tri(i, j)] = __builtin_fma(-lc[j], b[i], __builtin_fma(-lc[i], b[j], S[tri(i, j)]));
      S[tri(i, k)] = __builtin_fma(lc[i], htkk, b[i]);
    }
    pin(S);
    ODEF_SCHED_FENCE();
  });
  {
    double di[D];
    static_for<0, D>([&](auto kc) { di[decltype(kc)::value] = aget<LS + tri(decltype(kc)::value, decltype(kc)::value)>(); });
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int j = 0; j < i; ++j) S[tri(i, j)] = (S[tri(i, j)] * di[i]) * di[j];
      S[tri(i, i)] = __builtin_fma(S[tri(i, i)] * di[i], di[i], -di[i]);
    }
    pin(S);
    ODEF_SCHED_FENCE();
  }
  static_for<0, D>([&](auto kc) {
    constexpr int k = D - 1 - decltype(kc)::value;
    double lc[D];
    static_for<k + 1, D>([&](auto ic) { lc[decltype(ic)::value] = aget<LS + tri(decltype(ic)::value, k)>(); });
    double t[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double s = S[symidx(k, j)];
#pragma unroll
      for (int i = k + 1; i < D; ++i) s -= lc[i] * S[symidx(i, j)];
      t[j] = s;
    }
    double tk = t[k];
#pragma unroll
    for (int i = k + 1; i < D; ++i) tk -= lc[i] * t[i];
#pragma unroll
    for (int j = 0; j < D; ++j)
      if (j != k) S[symidx(k, j)] = t[j];
    S[tri(k, k)] = tk;
    pin(S);
    ODEF_SCHED_FENCE();
  });
}

// `n_hi`: wave-uniform upper bound of the record count of the lanes of this wavefront (fixed grid: n_save).
// ADAPT: per-trajectory record counts and step sizes (adaptive solve); otherwise everything about the grid is
// wave-uniform and comes from the host tables.
// The smoothed moments of time s+1 are CARRIED on chip (un-preconditioned, exactly the values stored): every record is
// read once and written once.  The record slot is wave-uniform: records are addressed through buffer descriptors
// (RowLoad / RowStore: row offset in a running SGPR, lane * 8 in voffset), no per-access VALU address arithmetic.
// (Rounds 1-2 walked 64-bit pointers: 273 v_lshl_add_u64 per step.)
template <int d, int q, bool ADAPT>
__device__ inline void smooth_lane_v2(const SmoothParams& P, long i0, unsigned lane, const LaneMem& xl, long n_hi) {
  constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  // AGPR-file slots: matrix, carried mean -- at the TOP of the file: should the compiler ever park a value of
  // its own in an AGPR it takes the lowest free one, and tests/test_build_hygiene.py fails the build if one of those reaches
  // the slots used here
  constexpr int MS = 128 - (TRI + D), VS = MS + TRI;
  static_assert(MS >= 0, "the AGPR file holds 128 doubles");
  agpr_file_claim();
  const long i = i0 + lane;
  const long n = ADAPT ? (long)P.nsaved[i] : P.n_save;
  const size_t N = (size_t)P.N;
  const PriorConsts& pc = P.pc;
  for (int w = 0; w < 2; ++w) {  // first and last record are copied (src/smoothing.jl:11)
    const long s = w == 0 ? 0 : n - 1;
    double c[TRI], m[D];
    const double* src = P.cov + ((size_t)s * TRI) * N + i;
#pragma unroll
    for (int k = 0; k < TRI; ++k) c[k] = src[(size_t)k * N];  // all loads in flight before the first store
#pragma unroll
    for (int k = 0; k < D; ++k) m[k] = P.mean[((size_t)s * D + k) * N + i];
    ODEF_SCHED_FENCE();
    double* dst = P.scov + ((size_t)s * TRI) * N + i;
#pragma unroll
    for (int k = 0; k < TRI; ++k) dst[(size_t)k * N] = c[k];
#pragma unroll
    for (int k = 0; k < D; ++k) P.smean[((size_t)s * D + k) * N + i] = m[k];
    // the last record starts the carried state (written on both passes: the second one stays)
    static_for<0, TRI>([&](auto kc) { aput<MS + decltype(kc)::value>(c[decltype(kc)::value]); });
    static_for<0, D>([&](auto kc) { aput<VS + decltype(kc)::value>(m[decltype(kc)::value]); });
  }
  bool nan_seen = false;
  for (long s = n_hi - 2; s >= 1; --s) {
    if constexpr (ADAPT) {
      if (s > n - 2) continue;  // this trajectory has fewer records: it joins at its own last one (the file holds it)
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
      h = uniform_load(P.hs + s);
      const GlobalTab tab{P.ptab + (size_t)uniform_load(P.tab_idx + s) * kTabStride};
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        pj[J] = tab[kTabPJ + J];
        pij[J] = tab[kTabPIJ + J];
      }
    }
    RowStore sm(P.smean + ((size_t)s * D) * N + i0, N, D, lane), sc(P.scov + ((size_t)s * TRI) * N + i0, N, TRI, lane);
    if (h == 0.0) {  // src/smoothing.jl:13-16
      static_for<0, D>([&](auto kc) { sm.put(aget<VS + decltype(kc)::value>()); });
      static_for<0, TRI>([&](auto kc) { sc.put(aget<MS + decltype(kc)::value>()); });
      continue;
    }
    const double sigma2 = P.diff[(size_t)(s + 1) * N + i];
    // x_i in preconditioned coordinates (src/smoothing.jl:23-24).  Every load of the step in flight first (raw values, no
    // arithmetic in between), then the scaling.
    double mt[D], B[TRI];
    {
      RowLoad lc(P.cov + ((size_t)s * TRI) * N + i0, N, TRI, lane), lm(P.mean + ((size_t)s * D) * N + i0, N, D, lane);
#pragma unroll
      for (int k = 0; k < TRI; ++k) B[k] = lc.get();
#pragma unroll
      for (int k = 0; k < D; ++k) mt[k] = lm.get();
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
      }
    pin(B);
    // delta = P m^s_+ - A m~   (src/smoothing.jl:38,44)
    double dl[D];
    static_for<0, D>([&](auto kc) { dl[decltype(kc)::value] = aget<VS + decltype(kc)::value>(); });
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int a = 0; a < d; ++a) {
        double t = mt[J * d + a];
#pragma unroll
        for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * mt[j * d + a];
        dl[J * d + a] = pj[J] * dl[J * d + a] - t;
      }
    pin(dl);
    ODEF_SCHED_FENCE();
    predict_cov_inplace<d, NB>(pc, B, sigma2);  // src/smoothing.jl:38
    pin(B);
    ODEF_SCHED_FENCE();
    double dinv[D];  // 1 / D_k of S^- = L D L' (0 for a zeroed column)
    ldl_packed_lane<D>(B, dinv);
    pin(B);
    pin(dinv);
    ODEF_SCHED_FENCE();
    // w = A' B^-1 delta ;  m^s = m + X w  (src/smoothing.jl:42-44): unit-lower forward, D^-1, unit-upper backward
#pragma unroll
    for (int k = 0; k < D; ++k) {
      double t = dl[k];
#pragma unroll
      for (int c = 0; c < k; ++c) t -= B[tri(k, c)] * dl[c];
      dl[k] = t;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) dl[k] *= dinv[k];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
      double t = dl[k];
#pragma unroll
      for (int c = k + 1; c < D; ++c) t -= B[tri(c, k)] * dl[c];
      dl[k] = t;
    }
#pragma unroll
    for (int K = NB - 1; K >= 1; --K)  // w = A' (.) in place: block K takes the still-untouched blocks j < K
#pragma unroll
      for (int b = 0; b < d; ++b)
#pragma unroll
        for (int j = 0; j < K; ++j) dl[K * d + b] += pc.At[j][K] * dl[j * d + b];
    // each entry of X is read once for the two mean components it feeds
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int c = 0; c <= a; ++c) {
        const double x = xl.get(tri(a, c));
        mt[a] += x * dl[c];
        if (c != a) mt[c] += x * dl[a];
      }
    static_for<0, D>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const double v = mt[k] * pij[k / d];  // un-precondition (src/smoothing.jl:26)
      nan_seen = nan_seen || !(v == v);
      sm.put(v);
      aput<VS + k>(v);
    });
    ODEF_SCHED_FENCE();
    // L out to the file, P S^s_+ P in -- entry by entry through the same slot, S[k] taking over the register of L[k]
    double S[TRI];
    static_for<0, TRI>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int a = [] { int r = 0; while ((r + 1) * (r + 2) / 2 <= k) ++r; return r; }();
      constexpr int b = k - a * (a + 1) / 2;
      S[k] = aget<MS + k>();
      aput<MS + k>(a == b ? dinv[a] : B[k]);  // (the diagonal slot takes 1 / D_k)
      S[k] *= pj[a / d] * pj[b / d];
    });
    pin(S);
    ODEF_SCHED_FENCE();
    // Z = B^-1 (S^s_+ - B) B^-1 ;  W = A' Z A
    two_sided_inverse_parked<D, MS>(S);
    pin(S);
    ODEF_SCHED_FENCE();
    congruence_At_inplace<d, NB>(pc, S);
    pin(S);
    ODEF_SCHED_FENCE();
    // S^s = X + X W X, TWO rows at a time: u = x_a W for both, then every column b of X is read once from LDS for the two
    // results it feeds (648 instead of 1 080 LDS values per step at D = 12); stored, and put into the file for the next step,
    // as it is produced
    auto rows = [&](auto a0c, auto nrc) {
      constexpr int a0 = decltype(a0c)::value, nr = decltype(nrc)::value;  // rows a0 .. a0 + nr - 1
      double u[nr][D];
#pragma unroll
      for (int r = 0; r < nr; ++r) {
        double xa[D];
#pragma unroll
        for (int c = 0; c < D; ++c) xa[c] = xl.get(symidx(a0 + r, c));
#pragma unroll
        for (int c = 0; c < D; ++c) {
          double t = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) t += xa[k] * S[symidx(k, c)];
          u[r][c] = t;
        }
      }
      RowStore st[2] = {sc.at_row(tri(a0, 0)), sc.at_row(tri(a0 + nr - 1, 0))};  // (nr <= 2: the second one is unused for nr = 1)
      static_for<0, a0 + nr>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        double col[D];
#pragma unroll
        for (int k = 0; k < D; ++k) col[k] = xl.get(symidx(k, b));
        static_for<0, nr>([&](auto rc) {
          constexpr int a = a0 + decltype(rc)::value;
          if constexpr (b <= a) {
            double t = col[a];
#pragma unroll
            for (int k = 0; k < D; ++k) t += u[a - a0][k] * col[k];
            const double v = t * (pij[a / d] * pij[b / d]);
            st[a - a0].put(v);
            aput<MS + tri(a, b)>(v);
          }
        });
      });
      ODEF_SCHED_FENCE();
    };
    static_for<0, D / 2>([&](auto pc_) { rows(std::integral_constant<int, 2 * decltype(pc_)::value>{}, std::integral_constant<int, 2>{}); });
    if constexpr (D % 2 == 1) rows(std::integral_constant<int, D - 1>{}, std::integral_constant<int, 1>{});
  }
  if (nan_seen) P.retcode[i] = 3;  // "NaNs after smoothing" (src/smoothing.jl:25)
}

}  // namespace odef
