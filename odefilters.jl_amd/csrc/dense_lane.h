// Dense output / saveat: posterior at arbitrary times (src/solution.jl:165-210).
// One lane per (trajectory, query time): locate the interval, `predict` from the left filter state
// with P(h1), and -- when smoothed and t < t_end -- one `smooth` against the right smoothed state
// with P(h2).  Reuses the filter's in-place predict and the smoother's step core (D <= 12).
#pragma once
#include "smooth_lane.h"

namespace odef {

struct DenseParams {
  PriorConsts pc;
  long N;
  long n_save;           // fixed: number of saves; adaptive: capacity
  int adaptive, smoothed;
  const double* tgrid;   // fixed: [n_save]
  const double* tsave;   // adaptive: [n_save][N]
  const int* nsaved;     // [N]
  const double* mean;    // filter records
  const double* cov;
  const double* diff;
  const double* smean;   // smoothed records (when smoothed)
  const double* scov;
  const double* tq;      // [n_q] query times
  long n_q;
  double* qmean;         // [n_q][D][N]
  double* qcov;          // [n_q][TRI][N]
};

template <int q>
__device__ inline void precond_from_h(double h, double (&pj)[q + 1], double (&pij)[q + 1]) {
  double val = precond_val<q>(h);
#pragma unroll
  for (int J = 0; J <= q; ++J) {
    pj[J] = val;
    pij[J] = 1.0 / val;
    val *= h;
  }
}

template <int d, int q>
__device__ inline void dense_lane(const DenseParams& P, long i, long jq, const LaneMem& xl) {
  constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  const size_t N = (size_t)P.N;
  const long n = P.adaptive ? (long)P.nsaved[i] : P.n_save;
  const double tval = P.tq[jq];
  auto tat = [&](long s) { return P.adaptive ? P.tsave[(size_t)s * N + i] : P.tgrid[s]; };
  // idx = number of grid points <= tval (Julia's 1-based `sum(t .<= tval)`), by bisection
  long lo = 0, hi = n;  // invariant: t[lo-1] <= tval < t[hi]  (with t[-1] = -inf, t[n] = +inf)
  while (lo < hi) {
    const long mid = (lo + hi) / 2;
    if (tat(mid) <= tval) lo = mid + 1;
    else hi = mid;
  }
  const long idx = lo;      // 1-based index of the left neighbour
  const long il = idx - 1;  // 0-based
  double* qm = P.qmean + ((size_t)jq * D) * N + i;
  double* qc = P.qcov + ((size_t)jq * TRI) * N + i;
  if (idx <= 0) {  // tval < t0: the reference throws "Invalid t<t0" (src/solution.jl:169-171)
    for (int k = 0; k < D; ++k) qm[(size_t)k * N] = __builtin_nan("");
    for (int k = 0; k < TRI; ++k) qc[(size_t)k * N] = __builtin_nan("");
    return;
  }
  const bool sm = P.smoothed != 0;
  if (tat(il) == tval) {  // src/solution.jl:172-176: exactly a stored time
    const double* m = (sm ? P.smean : P.mean) + ((size_t)il * D) * N + i;
    const double* c = (sm ? P.scov : P.cov) + ((size_t)il * TRI) * N + i;
    for (int k = 0; k < D; ++k) qm[(size_t)k * N] = m[(size_t)k * N];
    for (int k = 0; k < TRI; ++k) qc[(size_t)k * N] = c[(size_t)k * N];
    return;
  }
  // diffusions[min(idx, end)] (src/solution.jl:181): slot s holds the diffusion of step s-1 -> s
  const long sd = (idx < n - 1) ? idx : n - 1;
  const double sigma2 = P.diff[(size_t)sd * N + i];
  // extrapolate: goal_pred = P1^-1 predict(P1 prev, A, Qh)  (src/solution.jl:184-189)
  const double h1 = tval - tat(il);
  double pj1[NB], pij1[NB];
  precond_from_h<q>(h1, pj1, pij1);
  double mt[D], B[TRI];
#pragma unroll
  for (int k = 0; k < D; ++k) mt[k] = pj1[k / d] * P.mean[((size_t)il * D + k) * N + i];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) B[tri(a, b)] = P.cov[((size_t)il * TRI + tri(a, b)) * N + i] * (pj1[a / d] * pj1[b / d]);
  double mp[D];
#pragma unroll
  for (int J = 0; J < NB; ++J)
#pragma unroll
    for (int a = 0; a < d; ++a) {
      double t = mt[J * d + a];
#pragma unroll
      for (int j = J + 1; j < NB; ++j) t += P.pc.At[J][j] * mt[j * d + a];
      mp[J * d + a] = t;
    }
  predict_cov_inplace<d, NB>(P.pc, B, sigma2);
  if (!sm || il >= n - 1) {  // filter posterior, or beyond the last time (src/solution.jl:191-193)
#pragma unroll
    for (int k = 0; k < D; ++k) qm[(size_t)k * N] = pij1[k / d] * mp[k];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) qc[(size_t)tri(a, b) * N] = B[tri(a, b)] * (pij1[a / d] * pij1[b / d]);
    return;
  }
  // smooth against x_smooth[idx+1] with P(h2)  (src/solution.jl:199-209)
  const double h2 = tat(il + 1) - tval;
  double pj2[NB], pij2[NB];
  precond_from_h<q>(h2, pj2, pij2);
  double msn[D], Cs[TRI];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    mt[k] = pj2[k / d] * (pij1[k / d] * mp[k]);  // P2 * (P1^-1 * goal_pred)
    msn[k] = pj2[k / d] * P.smean[((size_t)(il + 1) * D + k) * N + i];
  }
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const double x = (B[tri(a, b)] * (pij1[a / d] * pij1[b / d])) * (pj2[a / d] * pj2[b / d]);
      B[tri(a, b)] = x;
      xl.set(tri(a, b), x);
      Cs[tri(a, b)] = P.scov[((size_t)(il + 1) * TRI + tri(a, b)) * N + i] * (pj2[a / d] * pj2[b / d]);
    }
  double mout[D];
  auto sink = [&](int k, double v) { qc[(size_t)k * N] = v; };
  rts_step_core<d, NB>(P.pc, pij2, mt, B, Cs, msn, sigma2, xl, mout, sink);
#pragma unroll
  for (int k = 0; k < D; ++k) qm[(size_t)k * N] = mout[k];
}

}  // namespace odef
