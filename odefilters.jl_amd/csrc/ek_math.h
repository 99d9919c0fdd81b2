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

namespace odef {

constexpr int MAXNB = 6;  // order <= 5

// (q+1)x(q+1) scalar tables of the Kronecker-structured prior A = At (x) I_d, Q = Qt (x) I_d,
// Q_L = QLt (x) I_d (src/priors.jl:7-59).  Filled on the host, passed by value (SGPRs).
struct PriorConsts {
  double At[MAXNB][MAXNB];
  double Qt[MAXNB][MAXNB];
  double QLt[MAXNB][MAXNB];
};

__host__ __device__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j
__host__ __device__ constexpr int symidx(int i, int j) { return i >= j ? tri(i, j) : tri(j, i); }

struct StepAux {
  double sigma2_local;   // cache.local_diffusion
  double sigma2_global;  // cache.global_diffusion
  double loglik;         // cache.log_likelihood of this step
  double eest_num_sq[1]; // unused placeholder (keeps the struct POD-extensible)
  int chol_fix;          // number of non-positive pivots met by the Cholesky (QR-fallback cases)
};

// In-place Cholesky of a packed symmetric matrix (lower).  A non-positive pivot means the
// reference's `cholesky!(check=false)` would report failure and fall back to a QR
// (src/filtering.jl:38-47); for a positive *semi*-definite matrix the equivalent factor is
// obtained by zeroing that column, which is what is done here (counted in `fixes`).
template <int D>
__device__ inline void chol_packed(double (&B)[D * (D + 1) / 2], int& fixes) {
#pragma unroll
  for (int k = 0; k < D; ++k) {
    double piv = B[tri(k, k)];
    const bool ok = piv > 0.0;
    const double lkk = ok ? sqrt(piv) : 0.0;
    const double inv = ok ? 1.0 / lkk : 0.0;
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

// preconditioner (src/preconditioning.jl:1-17): p[J] = h^(J-q-1/2) by the reference's running
// product starting from `pval = h^(-q-1/2)`; pinv = inv(Diagonal).
template <int NB>
__device__ inline void precond_tables(double h, double pval, double (&pj)[NB], double (&pij)[NB]) {
  double val = pval;
#pragma unroll
  for (int J = 0; J < NB; ++J) {
    pj[J] = val;
    pij[J] = 1.0 / val;
    val *= h;
  }
}

// B = A C A' + sigma2 * Q  on packed symmetric storage, A = At (x) I_d (block upper
// triangular), i.e. the Gram matrix of `_L = [A*L  sqrt(sigma2)*Q_L]` (src/filtering.jl:34-35).
// `C` comes in already preconditioned.
template <int d, int NB>
__device__ inline void predict_cov_gram(const PriorConsts& pc, const double (&C)[d * NB * (d * NB + 1) / 2],
                                        double sigma2, double (&B)[d * NB * (d * NB + 1) / 2]) {
  constexpr int D = d * NB;
  // T = A * C  (D x D)
  double T[D][D];
#pragma unroll
  for (int J = 0; J < NB; ++J)
#pragma unroll
    for (int a = 0; a < d; ++a)
#pragma unroll
      for (int k = 0; k < D; ++k) {
        double s = C[symidx(J * d + a, k)];
#pragma unroll
        for (int j = J + 1; j < NB; ++j) s += pc.At[J][j] * C[symidx(j * d + a, k)];
        T[J * d + a][k] = s;
      }
  // B = T * A' (lower triangle) + sigma2 * Q
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int b = 0; b < d; ++b) {
        const int j = K * d + b;
        if (j > i) continue;
        double s = T[i][j];
#pragma unroll
        for (int k = K + 1; k < NB; ++k) s += T[i][k * d + b] * pc.At[K][k];
        if ((i % d) == b) s += sigma2 * pc.Qt[i / d][K];
        B[tri(i, j)] = s;
      }
}

template <class RHS, int q, bool IS_EK1>
struct EKStep {
  static constexpr int d = RHS::d;
  static constexpr int NB = q + 1;
  static constexpr int D = d * NB;
  static constexpr int TRI = D * (D + 1) / 2;

  // One attempted step (src/perform_step.jl:27-76).  Inputs m, C: current filter state
  // (un-preconditioned, as `cache.x`).  Outputs m_out, C_out: `cache.x_filt` (un-preconditioned).
  // err_scale[r] = sqrt(diag(H (sigma2_local Q) H'))  (src/perform_step.jl:148-158).
  __device__ static inline void run(const PriorConsts& pc, const double* __restrict__ p, double h, double pval,
                                    bool fixed_diffusion, bool want_loglik, int success_iter, double prev_global,
                                    const double (&m)[D], const double (&C)[TRI], double (&m_out)[D],
                                    double (&C_out)[TRI], double (&err_scale)[d], StepAux& aux) {
    double pj[NB], pij[NB];
    precond_tables<NB>(h, pval, pj, pij);

    // x~ = P x  (src/perform_step.jl:36-38)
    double mt[D];
#pragma unroll
    for (int i = 0; i < D; ++i) mt[i] = pj[i / d] * m[i];
    double Ct[TRI];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) Ct[tri(i, j)] = (C[tri(i, j)] * pj[i / d]) * pj[j / d];

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
    double up[d], du[d], z[d];
#pragma unroll
    for (int a = 0; a < d; ++a) up[a] = pij[0] * mp[a];
    RHS::f(up, p, du);
#pragma unroll
    for (int a = 0; a < d; ++a) z[a] = pij[1] * mp[d + a] - du[a];
    // H = (E1 - J E0) PI  -> blocks H0 = -J*pi0, H1 = I*pi1 ; EK0: H0 = 0
    double H0[d][d];
    if constexpr (IS_EK1) {
      double Jm[d][d];
      RHS::jac(up, p, Jm);
#pragma unroll
      for (int r = 0; r < d; ++r)
#pragma unroll
        for (int a = 0; a < d; ++a) H0[r][a] = (0.0 - Jm[r][a]) * pij[0];
    } else {
#pragma unroll
      for (int r = 0; r < d; ++r)
#pragma unroll
        for (int a = 0; a < d; ++a) H0[r][a] = 0.0;
    }
    const double h1 = pij[1];  // H1 = h1 * I

    // M = H Q_L (d x 2d nonzero), W = M M' = H Q H'   (src/diffusions.jl:78)
    double W[d][d];
    {
      double M0[d][d];
      const double m1 = h1 * pc.QLt[1][1];
#pragma unroll
      for (int r = 0; r < d; ++r)
#pragma unroll
        for (int a = 0; a < d; ++a) M0[r][a] = H0[r][a] * pc.QLt[0][0] + (r == a ? h1 * pc.QLt[1][0] : 0.0);
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

    double sigma2_pred = 1.0;  // diffusion used inside predict_cov!
    if (!fixed_diffusion) {
      // DynamicDiffusion (src/diffusions.jl:72-80): sigma^2 = z' (H Q H')^-1 z / d
      double Winv[d][d];
      spd_inverse<d>(W, Winv, nullptr);
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < d; ++r) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < d; ++c) t += Winv[r][c] * z[c];
        s += z[r] * t;
      }
      sigma2_pred = s / d;
      aux.sigma2_local = sigma2_pred;
      aux.sigma2_global = sigma2_pred;
    }

    // predict_cov! (src/filtering.jl:33-48): Gram + Cholesky
    double Lp[TRI];
    predict_cov_gram<d, NB>(pc, Ct, sigma2_pred, Lp);
    chol_packed<D>(Lp, aux.chol_fix);

    // HL = H L^-  (d x 2d nonzero), S = HL HL'  (src/perform_step.jl:54,129)
    double HL[d][2 * d];
#pragma unroll
    for (int r = 0; r < d; ++r)
#pragma unroll
      for (int c = 0; c < 2 * d; ++c) {
        double s = 0.0;
#pragma unroll
        for (int k = c; k < 2 * d; ++k) {
          if (k < d) s += H0[r][k] * Lp[tri(k, c)];
          else if (k - d == r) s += h1 * Lp[tri(k, c)];
        }
        HL[r][c] = s;
      }
    double S[d][d], Sinv[d][d];
#pragma unroll
    for (int r = 0; r < d; ++r)
#pragma unroll
      for (int s = 0; s <= r; ++s) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < 2 * d; ++c) t += HL[r][c] * HL[s][c];
        S[r][s] = t;
        S[s][r] = t;
      }
    double logdetS = 0.0;
    spd_inverse<d>(S, Sinv, want_loglik ? &logdetS : nullptr);
    double zSz = 0.0;
#pragma unroll
    for (int r = 0; r < d; ++r) {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) t += Sinv[r][c] * z[c];
      zSz += z[r] * t;
    }
    // logpdf(measurement, 0) (src/perform_step.jl:66)
    aux.loglik = want_loglik ? -0.5 * (zSz + logdetS + d * 1.8378770664093453) : 0.0;

    if (fixed_diffusion) {
      // FixedDiffusion (src/diffusions.jl:11-36): running mean of z' S^-1 z / d
      const double diffusion_t = zSz / d;
      aux.sigma2_local = diffusion_t;
      aux.sigma2_global = (success_iter == 0) ? diffusion_t : prev_global + (diffusion_t - prev_global) / success_iter;
    }

    // error estimate scale (src/perform_step.jl:148-158)
#pragma unroll
    for (int r = 0; r < d; ++r) err_scale[r] = sqrt(aux.sigma2_local * W[r][r]);

    // update! (src/filtering.jl:79-91):  K = P_p H' S^-1 = L^- (HL)' S^-1
    double K[D][d];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double w[d];
#pragma unroll
      for (int r = 0; r < d; ++r) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 2 * d; ++c)
          if (c <= i) s += Lp[tri(i, c)] * HL[r][c];
        w[r] = s;
      }
#pragma unroll
      for (int r = 0; r < d; ++r) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < d; ++c) s += w[c] * Sinv[c][r];
        K[i][r] = s;
      }
    }
    // m = m_p + K (0 - z);  L = (I - K H) L^-  (only the first 2d columns change)
    double mf[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double s = mp[i];
#pragma unroll
      for (int r = 0; r < d; ++r) s += K[i][r] * (0.0 - z[r]);
      mf[i] = s;
    }
    double Lf[D][2 * d];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < 2 * d; ++c) {
        double s = (c <= i) ? Lp[tri(i, c)] : 0.0;
#pragma unroll
        for (int r = 0; r < d; ++r) s -= K[i][r] * HL[r][c];
        Lf[i][c] = s;
      }
    // Sigma_filt = L L' (src/squarerootmatrix.jl:16), then un-precondition (src/perform_step.jl:73-75)
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 2 * d; ++c) s += Lf[i][c] * Lf[j][c];
#pragma unroll
        for (int c = 2 * d; c <= j; ++c) s += Lp[tri(i, c)] * Lp[tri(j, c)];
        C_out[tri(i, j)] = (s * pij[i / d]) * pij[j / d];
      }
#pragma unroll
    for (int i = 0; i < D; ++i) m_out[i] = pij[i / d] * mf[i];
  }
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
