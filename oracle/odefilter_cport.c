/* CPU restatement (plain C, OpenMP over trajectories) of the reference's fixed-step filter
 * loop -- TEST INFRASTRUCTURE / CPU BASELINE ONLY, never linked into libodefilter_hip.so.
 *
 * Follows, operation by operation, what ProbNumDiffEq v0.1.5 computes per step
 * (src/perform_step.jl:27-76 with the dynamic-diffusion branch :40-54):
 *   x~ = P x (:36-38);  m^- = A m~ (filtering.jl:22-25);  measure! (perform_step.jl:95-132);
 *   sigma^2 = z'(H Q H')^-1 z / d (diffusions.jl:72-80);
 *   predict_cov!: _L = [A L~, sqrt(sigma^2) Q_L], Sigma^- = _L _L', Cholesky (filtering.jl:33-41);
 *   S = (H L^-)(H L^-)' (perform_step.jl:54);  K = Sigma^- H' S^-1, m = m^- - K z,
 *   L = (I - K H) L^- (filtering.jl:85-89);  Sigma = L L' (squarerootmatrix.jl:16);  x = P^-1 x (:73-75).
 * Dense D x D algebra as the reference does it (no Kronecker shortcuts), but allocation-free,
 * so it is a *stronger* CPU baseline than the allocating Julia path.
 * It is validated against the numpy oracle in tests/test_cport.py.
 */
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 24
#define MAXd 4

typedef void (*rhs_fn)(const double* u, const double* p, double* du, double* J);

static void rhs_fhn(const double* u, const double* p, double* du, double* J) {
  const double a = p[0], b = p[1], c = p[2];
  du[0] = c * (u[0] - u[0] * u[0] * u[0] / 3.0 + u[1]);
  du[1] = -(1.0 / c) * (u[0] - a - b * u[1]);
  J[0] = c * (1.0 - u[0] * u[0]); J[1] = c; J[2] = -(1.0 / c); J[3] = b / c;
}
static void rhs_lorenz(const double* u, const double* p, double* du, double* J) {
  const double s = p[0], r = p[1], b = p[2];
  du[0] = s * (u[1] - u[0]); du[1] = u[0] * (r - u[2]) - u[1]; du[2] = u[0] * u[1] - b * u[2];
  J[0] = -s; J[1] = s; J[2] = 0; J[3] = r - u[2]; J[4] = -1; J[5] = -u[0]; J[6] = u[1]; J[7] = u[0]; J[8] = -b;
}
static void rhs_lv(const double* u, const double* p, double* du, double* J) {
  const double a = p[0], b = p[1], c = p[2], dd = p[3];
  du[0] = a * u[0] - b * u[0] * u[1]; du[1] = -c * u[1] + dd * u[0] * u[1];
  J[0] = a - b * u[1]; J[1] = -b * u[0]; J[2] = dd * u[1]; J[3] = -c + dd * u[0];
}

static int chol(int n, double* A /* n x n row-major, lower in place */) {
  for (int j = 0; j < n; ++j) {
    double s = A[j * n + j];
    for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
    if (!(s > 0.0)) return -1;
    const double l = sqrt(s);
    A[j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double t = A[i * n + j];
      for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / l;
    }
    for (int i = 0; i < j; ++i) A[i * n + j] = 0.0;
  }
  return 0;
}

/* x = S^-1 b for SPD S (n <= MAXd) by Cholesky */
static void spd_solve(int n, const double* S, const double* b, double* x) {
  double L[MAXd * MAXd], y[MAXd];
  memcpy(L, S, sizeof(double) * n * n);
  chol(n, L);
  for (int i = 0; i < n; ++i) { double t = b[i]; for (int k = 0; k < i; ++k) t -= L[i * n + k] * y[k]; y[i] = t / L[i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double t = y[i]; for (int k = i + 1; k < n; ++k) t -= L[k * n + i] * x[k]; x[i] = t / L[i * n + i]; }
}

/* One trajectory, nsteps fixed steps.  m: [D] in/out, L: [D*D] factor in/out (row-major),
 * cov_out: [D*D] Sigma of the final state.  Returns number of failed Choleskys. */
static int filter_one(int rhs_id, int d, int q, int ek1, const double* A, const double* QL, const double* p,
                      const double* hs, const double* pvals, long nsteps, double* m, double* L, double* cov_out,
                      double* diffusions /* [nsteps] or NULL */) {
  const int D = d * (q + 1);
  rhs_fn f = rhs_id == 0 ? rhs_fhn : rhs_id == 1 ? rhs_lorenz : rhs_lv;
  double P[MAXD], PI[MAXD], mt[MAXD], mp[MAXD], Lt[MAXD * MAXD], AL[MAXD * MAXD], Sg[MAXD * MAXD], Lp[MAXD * MAXD];
  double H[MAXd * MAXD], HQ[MAXd * MAXD], HL[MAXd * MAXD], W[MAXd * MAXd], S[MAXd * MAXd], K[MAXD * MAXd], IKH[MAXD * MAXD];
  double z[MAXd], du[MAXd], J[MAXd * MAXd], up[MAXd], tmp[MAXd];
  int fails = 0;
  for (long n = 0; n < nsteps; ++n) {
    double val = pvals[n];
    for (int j = 0; j <= q; ++j) { for (int i = 0; i < d; ++i) { P[j * d + i] = val; PI[j * d + i] = 1.0 / val; } val *= hs[n]; }
    for (int i = 0; i < D; ++i) { mt[i] = P[i] * m[i]; for (int j = 0; j < D; ++j) Lt[i * D + j] = P[i] * L[i * D + j]; }
    for (int i = 0; i < D; ++i) { double t = 0; for (int k = 0; k < D; ++k) t += A[i * D + k] * mt[k]; mp[i] = t; }
    for (int a = 0; a < d; ++a) up[a] = PI[a] * mp[a];
    f(up, p, du, J);
    for (int a = 0; a < d; ++a) z[a] = PI[d + a] * mp[d + a] - du[a];
    memset(H, 0, sizeof(double) * d * D);
    for (int r = 0; r < d; ++r) {
      H[r * D + d + r] = PI[d + r];
      if (ek1) for (int a = 0; a < d; ++a) H[r * D + a] = (0.0 - J[r * d + a]) * PI[a];
    }
    /* sigma^2 */
    for (int r = 0; r < d; ++r) for (int c = 0; c < D; ++c) { double t = 0; for (int k = 0; k < D; ++k) t += H[r * D + k] * QL[k * D + c]; HQ[r * D + c] = t; }
    for (int r = 0; r < d; ++r) for (int s = 0; s < d; ++s) { double t = 0; for (int c = 0; c < D; ++c) t += HQ[r * D + c] * HQ[s * D + c]; W[r * d + s] = t; }
    spd_solve(d, W, z, tmp);
    double sig2 = 0; for (int r = 0; r < d; ++r) sig2 += z[r] * tmp[r];
    sig2 /= d;
    if (diffusions) diffusions[n] = sig2;
    const double sq = sqrt(sig2);
    /* predict_cov!: Gram of [A L~, sq Q_L], Cholesky */
    for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) { double t = 0; for (int k = 0; k < D; ++k) t += A[i * D + k] * Lt[k * D + j]; AL[i * D + j] = t; }
    for (int i = 0; i < D; ++i) for (int j = 0; j <= i; ++j) {
      double t = 0;
      for (int k = 0; k < D; ++k) t += AL[i * D + k] * AL[j * D + k] + (sq * QL[i * D + k]) * (sq * QL[j * D + k]);
      Sg[i * D + j] = t; Sg[j * D + i] = t;
    }
    memcpy(Lp, Sg, sizeof(double) * D * D);
    if (chol(D, Lp)) { ++fails; }
    /* Sigma^- = Lp Lp' (SRMatrix.mat, squarerootmatrix.jl:16) */
    for (int i = 0; i < D; ++i) for (int j = 0; j <= i; ++j) { double t = 0; for (int k = 0; k <= j; ++k) t += Lp[i * D + k] * Lp[j * D + k]; Sg[i * D + j] = t; Sg[j * D + i] = t; }
    for (int r = 0; r < d; ++r) for (int c = 0; c < D; ++c) { double t = 0; for (int k = 0; k < D; ++k) t += H[r * D + k] * Lp[k * D + c]; HL[r * D + c] = t; }
    for (int r = 0; r < d; ++r) for (int s = 0; s < d; ++s) { double t = 0; for (int c = 0; c < D; ++c) t += HL[r * D + c] * HL[s * D + c]; S[r * d + s] = t; }
    /* K = Sigma^- H' S^-1 (row by row) */
    for (int i = 0; i < D; ++i) {
      double w[MAXd];
      for (int r = 0; r < d; ++r) { double t = 0; for (int k = 0; k < D; ++k) t += Sg[i * D + k] * H[r * D + k]; w[r] = t; }
      spd_solve(d, S, w, tmp);
      for (int r = 0; r < d; ++r) K[i * d + r] = tmp[r];
    }
    for (int i = 0; i < D; ++i) { double t = mp[i]; for (int r = 0; r < d; ++r) t += K[i * d + r] * (0.0 - z[r]); mt[i] = t; }
    for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) { double t = (i == j) ? 1.0 : 0.0; for (int r = 0; r < d; ++r) t -= K[i * d + r] * H[r * D + j]; IKH[i * D + j] = t; }
    for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) { double t = 0; for (int k = 0; k < D; ++k) t += IKH[i * D + k] * Lp[k * D + j]; L[i * D + j] = PI[i] * t; }
    for (int i = 0; i < D; ++i) m[i] = PI[i] * mt[i];
  }
  for (int i = 0; i < D; ++i) for (int j = 0; j <= i; ++j) { double t = 0; for (int k = 0; k < D; ++k) t += L[i * D + k] * L[j * D + k]; cov_out[i * D + j] = t; cov_out[j * D + i] = t; }
  return fails;
}

/* Ensemble: m0 [N][D] (Taylor-initialised states), p shared.  Outputs mean_out [N][D], cov_out [N][D][D]. */
int cport_filter_fixed(int rhs_id, int d, int q, int ek1, long N, const double* A, const double* QL, const double* p,
                       const double* hs, const double* pvals, long nsteps, const double* m0, double* mean_out,
                       double* cov_out, int nthreads) {
  const int D = d * (q + 1);
  if (D > MAXD || d > MAXd) return -1;
  int fails = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static) reduction(+ : fails)
  for (long i = 0; i < N; ++i) {
    double m[MAXD], L[MAXD * MAXD];
    memcpy(m, m0 + i * D, sizeof(double) * D);
    memset(L, 0, sizeof(double) * D * D);
    fails += filter_one(rhs_id, d, q, ek1, A, QL, p, hs, pvals, nsteps, m, L, cov_out + i * D * D, 0);
    memcpy(mean_out + i * D, m, sizeof(double) * D);
  }
  return fails;
}

int cport_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
