// One extended-Kalman ODE-filter step for one trajectory held entirely in the registers of
// one lane (small state dimension D = d(q+1) <= ~18).  64 trajectories per wavefront.
//
// Follows src/perform_step.jl:27-93 of the reference step by step; what is different is
// only what is *stored* between steps: the reference keeps a `SquarerootMatrix` that holds
// both a factor and its Gram matrix `mat = S*S'` (src/squarerootmatrix.jl:10-16); here the
// carried quantity is the packed lower triangle of `mat` (78 doubles at D=12 instead of
// 144+144), while everything inside the step stays in square-root form exactly as in the
// reference: Cholesky factor of the predicted covariance (src/filtering.jl:33-41), Joseph
// update on that factor (src/filtering.jl:85-89), Gram of the updated factor (= what the
// reference's SRMatrix constructor computes, src/squarerootmatrix.jl:16).
#pragma once
#include "odef_platform.h"
#include "rhs.h"

// Scheduling fence: keeps the instruction scheduler from hoisting every load of a fully unrolled loop
// nest to the top (which costs hundreds of registers).  No code is emitted.
#ifdef ODEF_HOST_EMUL
#define ODEF_SCHED_FENCE()
#else
#define ODEF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

namespace odef {

constexpr int MAXNB = 6;  // order <= 5

// (q+1)x(q+1) scalar tables of the Kronecker-structured prior A = At (x) I_d, Q = Qt (x) I_d,
// Q_L = QLt (x) I_d (src/priors.jl:7-59).  Filled on the host, passed by value (SGPRs).
struct PriorConsts {
  double At[MAXNB][MAXNB];
  double Qt[MAXNB][MAXNB];
  double QLt[MAXNB][MAXNB];
};

// Compile-time loop: body(std::integral_constant<int, k>) for k = B .. E-1.  Used where `#pragma unroll`
// gives up ("unrolled size too large") and the fallback run-time loop would index register arrays dynamically.
template <int B, int E, class F>
__device__ inline void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

__host__ __device__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j
__host__ __device__ constexpr int symidx(int i, int j) { return i >= j ? tri(i, j) : tri(j, i); }

struct StepAux {
  double sigma2_local;   // cache.local_diffusion
  double sigma2_global;  // cache.global_diffusion
  double loglik;         // cache.log_likelihood of this step WITHOUT its -1/2 log det S term when `det` carries that (d <= 4)
  double det;            // d <= 4: sqrt(det S) (lane kernels: |prod R_kk|) or det S (row teams: prod D_k), see LogDetAcc; else 1
  double eest_num_sq[1]; // unused placeholder (keeps the struct POD-extensible)
  int chol_fix;          // number of non-positive pivots met by the Cholesky (QR-fallback cases)
};

// sum_n log(det_n) without a logarithm per step.  logpdf(measurement, 0) (src/perform_step.jl:66) needs log det S of every
// step; a double-precision log is ~70 instructions -- a tenth of the row-team filter's step -- and only the SUM over the
// steps is ever used (sol.log_likelihood).  So the determinants are multiplied up as (mantissa in [0.5, 1), binary exponent),
// five instructions per factor (v_frexp_mant_f64, v_frexp_exp_i32_f64), and the logarithm is taken once, when the
// trajectory's log-likelihood is stored.  A zero determinant (semi-definite S) gives -inf as log(0) did; the mantissa
// product of 1 024 factors is good to 1e-13, i.e. to 1e-13 absolute in a sum of magnitude 1e4.
struct LogDetAcc {
  double m;
  int e;
  __device__ inline void init() {
    m = 1.0;
    e = 0;
  }
  __device__ inline void mul(double det) {
    int ex;
    m *= frexp(det, &ex);
    e += ex;
    m = frexp(m, &ex);
    e += ex;
  }
  __device__ inline double log_value() const { return log(m) + (double)e * 0.69314718055994530942; }
};

// sqrt(x) and 1 / sqrt(x) together, for x > 0: v_rsq_f64 seed and two coupled Goldschmidt steps plus one residual correction each -- 14 instructions where a
// correctly rounded sqrt followed by a correctly rounded division takes 28 (the pivots of the three factorisations of a
// step are 12 such pairs, a tenth of its instructions).  Both results within an ulp (tools/rsqrt_accuracy.hip).
// x = 0 gives NaN (0 * inf in the first product), like any negative x: callers guard their pivots (x > 0) -- the
// Cholesky columns always did, the reflector norms and chol_small do since round 2.
__device__ inline void sqrt_and_rsqrt(double x, double& s, double& rs) {
#ifdef ODEF_HOST_EMUL
  // the device sequence itself, run on the host: the seed is 1/sqrt(x) cut to the 24 bits a v_rsq_f64 seed is good for,
  // the refinement is the same chain of fused multiply-adds, so tests/emul and the CPU sanitizers execute the arithmetic
  // the GPU executes (incl. x = 0 -> NaN)
  double y0;
  if (x == 0.0) {
    y0 = INFINITY;
  } else if (!(x > 0.0) || !(x <= 1.79769313486231570815e+308)) {
    y0 = (x > 0.0) ? 0.0 : NAN;
  } else {
    int e;
    const double mnt = std::frexp(1.0 / std::sqrt(x), &e);
    y0 = std::ldexp(std::nearbyint(std::ldexp(mnt, 24)), e - 24);
  }
#else
  const double y0 = __builtin_amdgcn_rsq(x);
#endif
  double g = x * y0;
  double h = 0.5 * y0;
  double r = __builtin_fma(-g, h, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-g, h, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double dd = __builtin_fma(-g, g, x);  // residual correction of the root
  g = __builtin_fma(dd, h, g);
  double y = h + h;
  const double e = __builtin_fma(-g, y, 1.0);  // and of its reciprocal
  y = __builtin_fma(e, y, y);
  s = g;
  rs = y;
}

// In-place Cholesky of a packed symmetric matrix (lower).  A non-positive pivot means the
// reference's `cholesky!(check=false)` would report failure and fall back to a QR
// (src/filtering.jl:38-47); for a positive *semi*-definite matrix the equivalent factor is
// obtained by zeroing that column, which is what is done here (counted in `fixes`).
// `dinv` receives the reciprocals of the diagonal (0 for a zeroed column): the smoother's two triangular transforms
// would otherwise divide by every pivot twice more.
template <int D>
__device__ inline void chol_packed(double (&B)[D * (D + 1) / 2], int& fixes, double (&dinv)[D]) {
#pragma unroll
  for (int k = 0; k < D; ++k) {
    double piv = B[tri(k, k)];
    const bool ok = piv > 0.0;
    double root, rroot;
    sqrt_and_rsqrt(piv, root, rroot);
    const double lkk = ok ? root : 0.0;
    const double inv = ok ? rroot : 0.0;
    dinv[k] = inv;
    fixes += ok ? 0 : 1;
    B[tri(k, k)] = lkk;
#pragma unroll
    for (int i = k + 1; i < D; ++i) B[tri(i, k)] *= inv;
#pragma unroll
    for (int j = k + 1; j < D; ++j) {
      const double ljk = B[tri(j, k)];
#pragma unroll
      for (int i = j; i < D; ++i) B[tri(i, j)] -= B[tri(i, k)] * ljk;
    }
  }
}
template <int D>
__device__ inline void chol_packed(double (&B)[D * (D + 1) / 2], int& fixes) {
  double dinv[D];
  chol_packed<D>(B, fixes, dinv);
}

// Cholesky-based inverse of a small SPD matrix; returns log(det) = 2*sum(log L_ii) if asked.
template <int n>
__device__ inline void spd_inverse(const double (&S)[n][n], double (&Sinv)[n][n], double* logdet) {
  double L[n][n];
#pragma unroll
  for (int j = 0; j < n; ++j) {
    double s = S[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
    const double ljj = sqrt(s);
    L[j][j] = ljj;
    const double inv = 1.0 / ljj;
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      double t = S[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
      L[i][j] = t * inv;
    }
  }
  // Linv (lower)
  double Li[n][n];
#pragma unroll
  for (int j = 0; j < n; ++j) {
    Li[j][j] = 1.0 / L[j][j];
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      double t = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) t -= L[i][k] * Li[k][j];
      Li[i][j] = t / L[i][i];
    }
  }
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double t = 0.0;
#pragma unroll
      for (int k = i; k < n; ++k) t += Li[k][i] * Li[k][j];
      Sinv[i][j] = t;
      Sinv[j][i] = t;
    }
  if (logdet) {
    double prod = 1.0;
    double acc = 0.0;
    if constexpr (n <= 4) {
#pragma unroll
      for (int i = 0; i < n; ++i) prod *= L[i][i];
      acc = log(prod);
    } else {
#pragma unroll
      for (int i = 0; i < n; ++i) acc += log(L[i][i]);
    }
    *logdet = 2.0 * acc;
  }
}

// ---- per-step preconditioner tables -------------------------------------------------------
// P(h) = diag(h^(j-q-1/2)) (x) I_d (src/preconditioning.jl:1-17), built by the reference's running
// product from pval = h^(-q-1/2).  Layout of one table (kTabStride doubles):
//   [0 .. MAXNB)            pj[J]   = P block value
//   [MAXNB .. 2 MAXNB)      pij[J]  = 1 / pj[J]                      (inv(::Diagonal))
//   [2 MAXNB + J*MAXNB + K] pp[J][K]   = pj[J]*pj[K]
//   [2 MAXNB + MAXNB^2 + J*MAXNB + K]  pipi[J][K] = pij[J]*pij[K]
// Fixed-step solves build one table per distinct h on the host (read through scalar loads:
// the pointer is wave-uniform); the adaptive kernel fills one per lane in registers.
constexpr int kTabPJ = 0, kTabPIJ = MAXNB, kTabPP = 2 * MAXNB, kTabPIPI = 2 * MAXNB + MAXNB * MAXNB;
constexpr int kTabStride = 2 * MAXNB + 2 * MAXNB * MAXNB;

template <int NB>
__host__ __device__ inline void precond_fill(double h, double pval, double* tab) {
  double val = pval;
#pragma unroll
  for (int J = 0; J < NB; ++J) {
    tab[kTabPJ + J] = val;
    tab[kTabPIJ + J] = 1.0 / val;
    val *= h;
  }
#pragma unroll
  for (int J = 0; J < NB; ++J)
#pragma unroll
    for (int K = 0; K < NB; ++K) {
      tab[kTabPP + J * MAXNB + K] = tab[kTabPJ + J] * tab[kTabPJ + K];
      tab[kTabPIPI + J * MAXNB + K] = tab[kTabPIJ + J] * tab[kTabPIJ + K];
    }
}

// 1 / x for a positive normal x: v_rcp_f64 seed and two Newton steps (5 instructions; libm's division is 10+)
__device__ inline double rcp_pos(double x) {
#ifdef ODEF_HOST_EMUL
  return 1.0 / x;
#else
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
#endif
}

// P(h) = diag(h^(j-q-1/2)) and its inverse for a step size that differs from trajectory to trajectory (adaptive steps;
// src/preconditioning.jl:1-17): running products from h^(-q-1/2) like the reference's, the reciprocals as a second
// running product (inv(::Diagonal) to within an ulp) -- no division, no libm pow: the table is rebuilt by all 16 lanes
// of a team at every attempted step.
template <int q, int NB, bool WITH_PRODUCTS = false>
__device__ inline void precond_table_fast(double h, double* tab) {
  double hq = 1.0;
#pragma unroll
  for (int k = 0; k < q; ++k) hq *= h;
  double sh, rsh;
  sqrt_and_rsqrt(h, sh, rsh);
  double val = rsh * rcp_pos(hq), ival = hq * sh;
  const double rh = rcp_pos(h);
#pragma unroll
  for (int J = 0; J < NB; ++J) {
    tab[kTabPJ + J] = val;
    tab[kTabPIJ + J] = ival;
    val *= h;
    ival *= rh;
  }
  if constexpr (WITH_PRODUCTS) {
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        tab[kTabPP + J * MAXNB + K] = tab[kTabPJ + J] * tab[kTabPJ + K];
        tab[kTabPIPI + J * MAXNB + K] = tab[kTabPIJ + J] * tab[kTabPIJ + K];
      }
  }
}

// X <- A X A' + sigma2 * Q  in place on packed symmetric storage, A = At (x) I_d block upper
// triangular: the Gram matrix of `_L = [A*L  sqrt(sigma2)*Q_L]` (src/filtering.jl:34-35).
// A = E_{NB-2} ... E_1 E_0 with E_J = I + sum_{j>J} At[J][j] e_J e_j' (block row J picks up the
// still-untouched block rows j > J), applied as successive symmetric congruences.
struct NoTick {
  __device__ inline void operator()() const {}
};
// `tick()` is called at regular points of the arithmetic; the lagged record sink of the filter (ek_lane.h) uses it to
// spread the stores of the previous record over the step.
// `T`: the scalar type of the covariance (double in every kernel; a host emulation of a multi-lane mapping may pass a vector type).
template <int d, int NB, class T, class Tick = NoTick>
__device__ inline void predict_cov_inplace(const PriorConsts& pc, T (&X)[d * NB * (d * NB + 1) / 2], T sigma2,
                                           Tick tick = Tick{}) {
  constexpr int D = d * NB;
#pragma unroll
  for (int J = 0; J + 1 < NB; ++J) {
    // W_JJ = X_JJ + sum_j a_j X_jJ from the old cells (full d x d block, not symmetric)
    T Wd[d][d];
#pragma unroll
    for (int a = 0; a < d; ++a)
#pragma unroll
      for (int b = 0; b < d; ++b) {
        T t = X[symidx(J * d + a, J * d + b)];
#pragma unroll
        for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * X[symidx(j * d + a, J * d + b)];
        Wd[a][b] = t;
      }
    tick();
    // off-diagonal blocks of block row J: X_Jk += sum_j a_j X_jk  (k != J)
#pragma unroll
    for (int a = 0; a < d; ++a) {
      tick();
#pragma unroll
      for (int c = 0; c < D; ++c) {
        if (c / d == J) continue;
        T t = X[symidx(J * d + a, c)];
#pragma unroll
        for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * X[symidx(j * d + a, c)];
        X[symidx(J * d + a, c)] = t;
      }
    }
    // Y_JJ = W_JJ + sum_j a_j W_Jj
#pragma unroll
    for (int a = 0; a < d; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        T t = Wd[a][b];
#pragma unroll
        for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * X[symidx(J * d + a, j * d + b)];
        X[tri(J * d + a, J * d + b)] = t;
      }
  }
#pragma unroll
  for (int J = 0; J < NB; ++J)
#pragma unroll
    for (int K = 0; K <= J; ++K)
#pragma unroll
      for (int a = 0; a < d; ++a) X[tri(J * d + a, K * d + a)] += sigma2 * pc.Qt[J][K];
}

// Cholesky of a small SPD matrix, lower factor L and the reciprocals of its diagonal.
template <int n>
__device__ inline void chol_small(const double (&S)[n][n], double (&L)[n][n], double (&Ldinv)[n]) {
#pragma unroll
  for (int j = 0; j < n; ++j) {
    double s = S[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
    const bool ok = s > 0.0;  // a non-positive pivot zeroes its column (the semi-definite rule of chol_packed)
    double ljj, inv;
    sqrt_and_rsqrt(ok ? s : 1.0, ljj, inv);
    ljj = ok ? ljj : 0.0;
    inv = ok ? inv : 0.0;
    L[j][j] = ljj;
    Ldinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      double t = S[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
      L[i][j] = t * inv;
    }
  }
}

// Global diffusion of the static models from this step's residual res_t = z' S^-1 z / d (`mode`: the C ABI's
// odef_diffusion value).  1: FixedDiffusion, running mean (src/diffusions.jl:26-35).  2: MAPFixedDiffusion, mode of
// the InverseGamma(1/2, 1/2) posterior rebuilt on-line from the previous estimate (src/diffusions.jl:46-68).
template <int d>
__host__ __device__ inline double static_diffusion_update(int mode, int success_iter, double prev, double res_t) {
  if (mode != 2) return (success_iter == 0) ? res_t : prev + (res_t - prev) / success_iter;
  const double alpha = 0.5, beta = 0.5;
  const int n_obs = success_iter + 1;
  if (success_iter == 0) return (beta + 0.5 * res_t) / (alpha + n_obs * d / 2.0 + 1.0);
  const double res_prev = (prev * (alpha + (n_obs - 1) * d / 2.0 + 1.0) - beta) * 2.0;
  const double res_sum_t = res_prev + res_t;
  return (beta + 0.5 * res_sum_t) / (alpha + n_obs * d / 2.0 + 1.0);
}

template <class RHS, int q, bool IS_EK1>
struct EKStep {
  static constexpr int d = RHS::d;
  static constexpr int NB = q + 1;
  static constexpr int D = d * NB;
  static constexpr int TRI = D * (D + 1) / 2;
  static constexpr int d2 = 2 * d;

  // One attempted step (src/perform_step.jl:27-76).  Inputs m, C: current filter state
  // (un-preconditioned, as `cache.x`; C = packed Sigma).  Outputs m_out, C_out: `cache.x_filt`.
  // err_scale[r] = sqrt(diag(H (sigma2_local Q) H'))  (src/perform_step.jl:148-158).
  // `tab`: preconditioner table of this step's h (see precond_fill).
  //
  // Square-root structure inside the step: the predicted covariance is Cholesky-factorised
  // (src/filtering.jl:36) -- only its first 2d columns are needed because H = (E1 - J E0) P^-1
  // has support on the first two derivative blocks -- S = (H L)(H L)' is formed from the factor
  // (src/perform_step.jl:54), and the Joseph update (I - K H) L (src/filtering.jl:89) is carried
  // out as the orthogonal transformation that triangularises (H L)': with (H L1)' = Q [R; 0],
  //   K = (L1 Q)[:, :d] R^-T,   Sigma_filt = (L1 Q)[:, d:2d] (L1 Q)[:, d:2d]' + Schur complement,
  // which is the Gram matrix of (I - K H) L without ever forming K H.
  // `sink.mean(v)` / `sink.cov(v)` receive the un-preconditioned results in storage order as soon as each
  // value exists; `sink.tick()` is called at ~110 points spread evenly over the arithmetic of the step (never
  // inside a run-time branch), so that a sink can spread its stores over the step (LaggedSink, ek_lane.h).
  template <class Sink, class Tab>
  __device__ static inline void run(const PriorConsts& pc, const double* __restrict__ p, const Tab& tab,
                                    int fixed_diffusion, bool want_loglik, int success_iter, double prev_global,
                                    const double (&m)[D], const double (&C)[TRI], double (&m_out)[D],
                                    double (&C_out)[TRI], double (&err_scale)[d], StepAux& aux, Sink& sink) {
    // x~ = P x  (src/perform_step.jl:36-38)
    double mt[D];
#pragma unroll
    for (int i = 0; i < D; ++i) mt[i] = tab[kTabPJ + i / d] * m[i];
    double X[TRI];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if (i % 2 == 1) sink.tick();
#pragma unroll
      for (int j = 0; j <= i; ++j) X[tri(i, j)] = C[tri(i, j)] * tab[kTabPP + (i / d) * MAXNB + (j / d)];
    }

    // predict mean (src/filtering.jl:22-25)
    double mp[D];
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int a = 0; a < d; ++a) {
        double s = mt[J * d + a];
#pragma unroll
        for (int j = J + 1; j < NB; ++j) s += pc.At[J][j] * mt[j * d + a];
        mp[J * d + a] = s;
      }

    // measure! (src/perform_step.jl:95-132)
    const double pi0 = tab[kTabPIJ + 0], pi1 = tab[kTabPIJ + 1];
    double up[d], du[d], z[d];
#pragma unroll
    for (int a = 0; a < d; ++a) up[a] = pi0 * mp[a];
    sink.tick();
    RHS::f(up, p, du);
#pragma unroll
    for (int a = 0; a < d; ++a) z[a] = pi1 * mp[d + a] - du[a];
    sink.tick();
    // H = (E1 - J E0) PI  -> blocks H0 = -J*pi0, H1 = I*pi1 ; EK0: H0 = 0
    double H0[d][d];
    if constexpr (IS_EK1) {
      double Jm[d][d];
      rhs_jacobian<RHS>(up, p, Jm);  // f.jac, else forward-mode AD (src/perform_step.jl:116-121)
#pragma unroll
      for (int r = 0; r < d; ++r)
#pragma unroll
        for (int a = 0; a < d; ++a) H0[r][a] = (0.0 - Jm[r][a]) * pi0;
    } else {
#pragma unroll
      for (int r = 0; r < d; ++r)
#pragma unroll
        for (int a = 0; a < d; ++a) H0[r][a] = 0.0;
    }
    const double h1 = pi1;  // H1 = h1 * I
    sink.tick();

    // M = H Q_L (d x 2d nonzero), W = M M' = H Q H'   (src/diffusions.jl:78)
    double W[d][d];
    {
      double M0[d][d];
      const double m1 = h1 * pc.QLt[1][1];
#pragma unroll
      for (int r = 0; r < d; ++r)
#pragma unroll
        for (int a = 0; a < d; ++a) {
          double t = (r == a) ? h1 * pc.QLt[1][0] : 0.0;
          if constexpr (IS_EK1) t += H0[r][a] * pc.QLt[0][0];
          M0[r][a] = t;
        }
#pragma unroll
      for (int r = 0; r < d; ++r)
#pragma unroll
        for (int s = 0; s <= r; ++s) {
          double t = (r == s) ? m1 * m1 : 0.0;
#pragma unroll
          for (int a = 0; a < d; ++a) t += M0[r][a] * M0[s][a];
          W[r][s] = t;
          W[s][r] = t;
        }
    }

    sink.tick();
    double sigma2_pred = 1.0;  // diffusion used inside predict_cov!
    if (!fixed_diffusion) {
      // DynamicDiffusion (src/diffusions.jl:72-80): sigma^2 = z' (H Q H')^-1 z / d = |Lw^-1 z|^2 / d
      double Lw[d][d], Lwi[d];
      chol_small<d>(W, Lw, Lwi);
      double s = 0.0, yw[d];
#pragma unroll
      for (int r = 0; r < d; ++r) {
        double t = z[r];
#pragma unroll
        for (int c = 0; c < r; ++c) t -= Lw[r][c] * yw[c];
        yw[r] = t * Lwi[r];
        s += yw[r] * yw[r];
      }
      sigma2_pred = s / d;
      aux.sigma2_local = sigma2_pred;
      aux.sigma2_global = sigma2_pred;
    }

    // predict_cov! (src/filtering.jl:33-41): Gram matrix, then Cholesky -- first 2d columns,
    // right-looking, so that X[i>=2d][j>=2d] ends as the Schur complement L2 L2'.
    sink.tick();
    predict_cov_inplace<d, NB>(pc, X, sigma2_pred, [&]() { sink.tick(); });
#pragma unroll
    for (int k = 0; k < d2; ++k) {
      sink.tick();
      const double piv = X[tri(k, k)];
      const bool ok = piv > 0.0;  // a failing pivot is the reference's QR-fallback case (src/filtering.jl:38-47)
      double root, rroot;
      sqrt_and_rsqrt(piv, root, rroot);
      const double lkk = ok ? root : 0.0;
      const double inv = ok ? rroot : 0.0;
      aux.chol_fix += ok ? 0 : 1;
      X[tri(k, k)] = lkk;
#pragma unroll
      for (int i = k + 1; i < D; ++i) X[tri(i, k)] *= inv;
#pragma unroll
      for (int j = k + 1; j < D; ++j) {
        if ((j - k) % 4 == 0) sink.tick();
        const double ljk = X[tri(j, k)];
#pragma unroll
        for (int i = j; i < D; ++i) X[tri(i, j)] -= X[tri(i, k)] * ljk;
      }
    }

    // G = (H L1)' (2d x d):  G[c][r] = sum_{k>=c} H[r][k] L[k][c]
    double G[d2][d];
#pragma unroll
    for (int c = 0; c < d2; ++c) {
      if (c % 2 == 0) sink.tick();
#pragma unroll
      for (int r = 0; r < d; ++r) {
        double s = 0.0;
        if constexpr (IS_EK1) {
#pragma unroll
          for (int k = c; k < d; ++k) s += H0[r][k] * X[tri(k, c)];
        }
        if (d + r >= c) s += h1 * X[tri(d + r, c)];
        G[c][r] = s;
      }
    }
    // Householder QR of G: G = Q [R; 0];  S = G'G = R'R  (measurement covariance, src/perform_step.jl:54)
    double hv[d][d2], hbeta[d], R[d][d], Rdinv[d];
#pragma unroll
    for (int k = 0; k < d; ++k) {
      sink.tick();
      double nrm2 = 0.0;
#pragma unroll
      for (int i = k; i < d2; ++i) nrm2 += G[i][k] * G[i][k];
      // a zero column (H L1 = 0: zero predicted covariance in the measured directions) has no reflector: R_kk = 0,
      // its reciprocal is taken as 0 like the reciprocal of a zero Cholesky pivot, beta = 0 below
      const bool nz = nrm2 > 0.0;
      double nrm, rnrm;
      sqrt_and_rsqrt(nz ? nrm2 : 1.0, nrm, rnrm);
      nrm = nz ? nrm : 0.0;
      rnrm = nz ? rnrm : 0.0;
      const double x0 = G[k][k];
      const double alpha = (x0 >= 0.0) ? -nrm : nrm;
      Rdinv[k] = (x0 >= 0.0) ? -rnrm : rnrm;  // 1 / R[k][k]
      const double v0 = x0 - alpha;
      const double vtv = nrm2 - x0 * x0 + v0 * v0;
      const double beta = (vtv > 0.0) ? 2.0 / vtv : 0.0;
      hbeta[k] = beta;
      hv[k][k] = v0;
#pragma unroll
      for (int i = k + 1; i < d2; ++i) hv[k][i] = G[i][k];
      R[k][k] = alpha;
      sink.tick();
#pragma unroll
      for (int c = k + 1; c < d; ++c) {
        double s = v0 * G[k][c];
#pragma unroll
        for (int i = k + 1; i < d2; ++i) s += hv[k][i] * G[i][c];
        s *= beta;
        G[k][c] -= s * v0;
#pragma unroll
        for (int i = k + 1; i < d2; ++i) G[i][c] -= s * hv[k][i];
        R[k][c] = G[k][c];
      }
    }
    // y = R^-T z ;  z' S^-1 z = y'y ;  log det S = 2 sum log |R_kk|
    double y[d], zSz = 0.0, detprod = 1.0, logacc = 0.0;
#pragma unroll
    for (int r = 0; r < d; ++r) {
      sink.tick();
      double t = z[r];
#pragma unroll
      for (int c = 0; c < r; ++c) t -= R[c][r] * y[c];
      y[r] = t * Rdinv[r];
      zSz += y[r] * y[r];
      if constexpr (d <= 4) detprod *= R[r][r];
      else if (want_loglik) logacc += log(fabs(R[r][r]));
    }
    if (want_loglik) {
      // logpdf(measurement, 0) (src/perform_step.jl:66); d <= 4: log det S = 2 log |prod R_kk| is left to the caller's LogDetAcc
      aux.loglik = -0.5 * (zSz + 2.0 * logacc + d * 1.8378770664093453);
    } else {
      aux.loglik = 0.0;
    }
    aux.det = (d <= 4) ? fabs(detprod) : 1.0;

    if (fixed_diffusion) {
      // FixedDiffusion (src/diffusions.jl:11-36): running mean of z' S^-1 z / d;  MAPFixedDiffusion (:46-68)
      const double diffusion_t = zSz / d;
      aux.sigma2_local = diffusion_t;
      aux.sigma2_global = static_diffusion_update<d>(fixed_diffusion, success_iter, prev_global, diffusion_t);
    }

    // error estimate scale (src/perform_step.jl:148-158)
    sink.tick();
#pragma unroll
    for (int r = 0; r < d; ++r) err_scale[r] = sqrt(aux.sigma2_local * W[r][r]);
    sink.tick();

    // update! (src/filtering.jl:79-91): rows of L1 times Q, Q = H_1 ... H_d.  Only d + 1 combinations of the columns of Q
    // are needed -- Q [y; 0] for the mean (K z = (L1 Q)[:, :d] y) and Q e_{d+r} for Z = (L1 Q)[:, d:2d] -- so the
    // reflectors are applied to those d + 1 vectors once instead of to every one of the D rows of L1.
    double qy[d2], qz[d][d2];
#pragma unroll
    for (int c = 0; c < d2; ++c) qy[c] = (c < d) ? y[c < d ? c : 0] : 0.0;
#pragma unroll
    for (int r = 0; r < d; ++r)
#pragma unroll
      for (int c = 0; c < d2; ++c) qz[r][c] = (c == d + r) ? 1.0 : 0.0;
#pragma unroll
    for (int k = d - 1; k >= 0; --k) {  // Q x = H_1 (H_2 (... H_d x))
      sink.tick();
      {
        double s = 0.0;
#pragma unroll
        for (int c = k; c < d2; ++c) s += hv[k][c] * qy[c];
        s *= hbeta[k];
#pragma unroll
        for (int c = k; c < d2; ++c) qy[c] -= s * hv[k][c];
      }
#pragma unroll
      for (int r = 0; r < d; ++r) {
        double s = 0.0;
#pragma unroll
        for (int c = k; c < d2; ++c) s += hv[k][c] * qz[r][c];
        s *= hbeta[k];
#pragma unroll
        for (int c = k; c < d2; ++c) qz[r][c] -= s * hv[k][c];
      }
    }
    double Zp[D][d];
#pragma unroll
    for (int l = 0; l < D; ++l) {
      sink.tick();
      // m = m_p + K (0 - z)
      double t = mp[l];
#pragma unroll
      for (int c = 0; c < d2; ++c)
        if (c <= l) t -= X[tri(l, c)] * qy[c];
      m_out[l] = tab[kTabPIJ + l / d] * t;  // un-precondition (src/perform_step.jl:75)
      sink.mean(m_out[l]);
      sink.tick();
#pragma unroll
      for (int r = 0; r < d; ++r) {
        double z2 = 0.0;
#pragma unroll
        for (int c = 0; c < d2; ++c)
          if (c <= l) z2 += X[tri(l, c)] * qz[r][c];
        Zp[l][r] = z2;
      }
      sink.tick();
    }
    // Sigma_filt = Zp Zp' + Schur, then un-precondition (src/perform_step.jl:73-75)
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        if (tri(i, j) % 4 == 0) sink.tick();
        double s = (j >= d2) ? X[tri(i, j)] : 0.0;
#pragma unroll
        for (int r = 0; r < d; ++r) s += Zp[i][r] * Zp[j][r];
        C_out[tri(i, j)] = s * tab[kTabPIPI + (i / d) * MAXNB + (j / d)];
        sink.cov(C_out[tri(i, j)]);
      }
  }
};

struct NoSink {
  __device__ inline void mean(double) {}
  __device__ inline void cov(double) {}
  __device__ inline void tick() {}
};

// Taylor-mode initialisation (src/state_initialization.jl:2-53): m0 = [u0; u'(t0); ...; u^(q)(t0)],
// Sigma0 = 0 (the q+1 exact `condition_on!` calls leave a zero covariance, test/solution.jl:38-41).
template <class RHS, int q>
__device__ inline void taylor_init(const double (&u0)[RHS::d], const double* __restrict__ p,
                                   double (&m0)[RHS::d * (q + 1)]) {
  constexpr int d = RHS::d, NB = q + 1;
  double coef[d][NB];
#pragma unroll
  for (int a = 0; a < d; ++a) {
    coef[a][0] = u0[a];
#pragma unroll
    for (int k = 1; k < NB; ++k) coef[a][k] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < q; ++k) {
    Jet<NB> u[d], fu[d];
#pragma unroll
    for (int a = 0; a < d; ++a)
#pragma unroll
      for (int c = 0; c < NB; ++c) u[a].c[c] = coef[a][c];
    RHS::f(u, p, fu);
#pragma unroll
    for (int a = 0; a < d; ++a) coef[a][k + 1] = fu[a].c[k] / (k + 1);
  }
  double fact = 1.0;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    if (k > 0) fact *= k;
#pragma unroll
    for (int a = 0; a < d; ++a) m0[k * d + a] = coef[a][k] * fact;
  }
}

}  // namespace odef
