// Dense output / saveat for 12 < D <= 32 (src/solution.jl:165-210) on the row-per-lane teams of smooth_rows.h: one team of 16 / 32
// lanes per (trajectory, query time) item -- `predict` from the left filter state with P(h1) (rows_predict_phase) and, for the
// smoothed posterior inside the grid, one step of the predicted state against the right smoothed state with P(h2)
// (rows_predict_phase + rows_gain_phase).  Same semantics as dense_lane.h (D <= 12) and dense_mfma.h (D = 168).
#pragma once
#include "dense_lane.h"
#include "smooth_rows.h"

namespace odef {

template <int d, int q, int TEAM>
__device__ inline void dense_rows_lane(const DenseParams& P, long i, long jq, int tid, double* __restrict__ ws, RowState<d*(q + 1)>* st) {
  constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  const size_t N = (size_t)P.N;
  const long n = P.adaptive ? (long)P.nsaved[i] : P.n_save;
  const double tval = P.tq[jq];
  const PriorConsts& pc = P.pc;
  const Team<TEAM> t{tid};
  (void)t;
  auto tat = [&](long s) { return P.adaptive ? P.tsave[(size_t)s * N + i] : P.tgrid[s]; };
  long lo = 0, hi = n;  // idx = number of grid points <= tval, by bisection (every lane the same walk)
  while (lo < hi) {
    const long mid = (lo + hi) / 2;
    if (tat(mid) <= tval) lo = mid + 1;
    else hi = mid;
  }
  const long idx = lo, il = idx - 1;
  double* qm = P.qmean + ((size_t)jq * D) * N + i;
  double* qc = P.qcov + ((size_t)jq * TRI) * N + i;
  const bool sm = P.smoothed != 0;
  if (idx <= 0 || tat(il) == tval) {  // before t0: NaN record (the reference throws); exactly a stored time: the record itself
    ODEF_ROWS_PHASE(
      if (r < D) {
        const bool bad = idx <= 0;
        const double* m = (sm ? P.smean : P.mean) + ((size_t)(bad ? 0 : il) * D) * N + i;
        const double* c = (sm ? P.scov : P.cov) + ((size_t)(bad ? 0 : il) * TRI) * N + i;
        qm[(size_t)r * N] = bad ? __builtin_nan("") : m[(size_t)r * N];
_Pragma("unroll")
        for (int cc = 0; cc < D; ++cc)
          if (cc <= r) qc[(size_t)tri(r, cc) * N] = bad ? __builtin_nan("") : c[(size_t)tri(r, cc) * N];
      }
    )
    return;
  }
  const long sd = (idx < n - 1) ? idx : n - 1;  // diffusions[min(idx, end)] (src/solution.jl:181)
  const double sigma2 = P.diff[(size_t)sd * N + i];
  const double h1 = tval - tat(il);
  double pj1[NB], pij1[NB];
  precond_from_h<q>(h1, pj1, pij1);
  // lane constants, the left filter state preconditioned with P(h1), nothing to smooth against yet
  ODEF_ROWS_PHASE(
    if (r < D) {
_Pragma("unroll")
      for (int J = 0; J < NB; ++J) {
        if (J == r / d) {
_Pragma("unroll")
          for (int j = 0; j < NB; ++j) {
            L.atr[j] = pc.At[J][j];
            L.qtr[j] = pc.Qt[J][j];
          }
        }
      }
      double pj_r = pj1[0], pij_r = pij1[0];
_Pragma("unroll")
      for (int J = 1; J < NB; ++J) {
        pj_r = (r / d == J) ? pj1[J] : pj_r;
        pij_r = (r / d == J) ? pij1[J] : pij_r;
      }
      L.pj = pj_r;
      L.pij = pij_r;
      L.mf = pj_r * P.mean[((size_t)il * D + r) * N + i];
      L.ms = 0.0;
_Pragma("unroll")
      for (int c = 0; c < D; ++c) {
        L.xr[c] = P.cov[((size_t)il * TRI + symidx(r, c)) * N + i] * (pj_r * pj1[c / d]);
        L.csr[c] = 0.0;
      }
    }
  )
  rows_predict_phase<d, q, TEAM>(pc, pj1, sigma2, tid, ws, st);  // L.lr = row of A X A' + sigma2 Q, L.mpred = (A m~)_r
  if (!sm || il >= n - 1) {  // filter posterior, or beyond the last time (src/solution.jl:191-193)
    ODEF_ROWS_PHASE(
      if (r < D) {
        qm[(size_t)r * N] = L.pij * L.mpred;
_Pragma("unroll")
        for (int c = 0; c < D; ++c)
          if (c <= r) qc[(size_t)tri(r, c) * N] = L.lr[c] * (L.pij * pij1[c / d]);
      }
    )
    return;
  }
  // smooth against x_smooth[idx+1] with P(h2)  (src/solution.jl:199-209): the predicted state plays the filter state
  const double h2 = tat(il + 1) - tval;
  double pj2[NB], pij2[NB], f12[NB];
  precond_from_h<q>(h2, pj2, pij2);
#pragma unroll
  for (int J = 0; J < NB; ++J) f12[J] = pj2[J] * pij1[J];
  ODEF_ROWS_PHASE(
    if (r < D) {
      double pj_r = pj2[0], pij_r = pij2[0], f_r = f12[0];
_Pragma("unroll")
      for (int J = 1; J < NB; ++J) {
        pj_r = (r / d == J) ? pj2[J] : pj_r;
        pij_r = (r / d == J) ? pij2[J] : pij_r;
        f_r = (r / d == J) ? f12[J] : f_r;
      }
      L.mf = pj_r * (L.pij * L.mpred);  // P2 (P1^-1 goal_pred)
      L.pj = pj_r;
      L.pij = pij_r;
      L.ms = P.smean[((size_t)(il + 1) * D + r) * N + i];
_Pragma("unroll")
      for (int c = 0; c < D; ++c) {
        L.xr[c] = L.lr[c] * (f_r * f12[c / d]);
        L.csr[c] = P.scov[((size_t)(il + 1) * TRI + symidx(r, c)) * N + i];
      }
    }
  )
  rows_predict_phase<d, q, TEAM>(pc, pj2, sigma2, tid, ws, st);
  rows_gain_phase<d, q, TEAM>(pij2, tid, ws, st);
  ODEF_ROWS_PHASE(
    if (r < D) {
      qm[(size_t)r * N] = L.ms;
_Pragma("unroll")
      for (int c = 0; c < D; ++c)
        if (c <= r) qc[(size_t)tri(r, c) * N] = L.csr[c];
    }
  )
}

}  // namespace odef
