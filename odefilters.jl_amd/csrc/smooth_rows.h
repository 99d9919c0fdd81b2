// RTS smoother, row-per-lane teams: TEAM lanes (16 or 32, TEAM >= D) own one trajectory; lane r
// keeps ROW r of every matrix in registers and only what other lanes must see goes through the
// team's LDS workspace.  4 (TEAM = 16) trajectories per wavefront, no s_barrier: LDS operations
// of one wave complete in issue order, `sync` is a wave-scope fence.
//
// Per step (src/smoothing.jl:31-63, src/filtering.jl:136-154), in preconditioned coordinates:
//   Y = S A' (own row) -> LDS;  B = A Y + sigma2 Q (row r);  M = S^s_+ - B (row r) -> LDS
//   B = L D L'  (unit lower L, row r in registers; column k exchanged through LDS; no square roots)
//   G = Y B^-1  (own row: forward, scale by 1/D, backward; L read from LDS as broadcasts)
//   m^s = m + G (m^s_+ - A m);   S^s = S + G M G'  (row r; M and G rows read from LDS)
// The textbook form S + G (S^s_+ - S^-) G' is the identity the reference's own test asserts for its
// stacked-QR Joseph form (test/filtering.jl:113); both agree to the oracle's rounding noise.
//
// The body is written as PHASES separated by team syncs.  On the device every lane runs every phase
// on its own `RowState`; the host emulation (tests/emul) runs each phase for all lanes in turn over
// an array of RowStates -- same source, same arithmetic.
#pragma once
#include "ek_lane.h"
#include "team.h"

namespace odef {

template <int d, int NB>
struct RowsWs {  // LDS workspace per team (doubles)
  static constexpr int D = d * NB, LD = team_ld(D);
  static constexpr int YL = 0;             // D x LD : Y, later the unit-lower factor L
  static constexpr int MM = YL + D * LD;   // D x LD : M = S^s_+ - S^-, later the lower part of S^s for symmetrisation
  static constexpr int GG = MM + D * LD;   // D x LD : G
  static constexpr int COL = GG + D * LD;  // 2 x D  : column exchange of the factorisation (double buffered)
  static constexpr int DINV = COL + 2 * D; // D      : 1 / D_k
  static constexpr int VMT = DINV + D;     // D      : m~
  static constexpr int VDL = VMT + D;      // D      : m^s_+ - m^-
  static constexpr int used = VDL + D;
  static constexpr int size = used + ((8 - used % 32) + 32) % 32;  // teams of a wave 16 banks apart
};

template <int D>
struct RowState {
  double xr[D];   // row r of P S P (filter covariance of time i)
  double yr[D];   // row r of Y, then of G
  double lr[D];   // row r of B, then of the unit-lower factor (columns <= r)
  double csr[D];  // row r of the carried smoothed covariance (un-preconditioned)
  double pj, pij; // own preconditioner entry and its inverse
  double mf, ms;  // own component of the filter mean (preconditioned) / carried smoothed mean (un-preconditioned)
  double mpred;   // own component of the predicted mean A m~ (preconditioned)
  double atr[MAXNB], qtr[MAXNB];  // row (r / d) of At and Qt: lane constants
};

#ifdef ODEF_HOST_EMUL
#define ODEF_ROWS_PHASE(...)                                   \
  for (int lane_ = 0; lane_ < TEAM; ++lane_) {                 \
    RowState<D>& L = st[lane_];                                \
    const int r = lane_;                                       \
    (void)L; (void)r;                                          \
    __VA_ARGS__                                                \
  }
#else
#define ODEF_ROWS_PHASE(...)                                   \
  {                                                            \
    RowState<D>& L = st[0];                                    \
    const int r = tid;                                         \
    (void)L; (void)r;                                          \
    __VA_ARGS__                                                \
  }                                                            \
  t.sync();
#endif

// L D L' of the symmetric matrix whose row r sits in L.lr (columns <= r referenced), right-looking, column k exchanged through
// LDS (COL: 2 x D, DINV: D).  Afterwards L.lr[c], c < r, is the unit-lower factor, L.lr[r] the pivot D_r, DINV[k] = 1 / D_k
// (semi-definite rule: 0 for a non-positive pivot, which zeroes its column).
template <int D, int TEAM>
__device__ inline void rows_ldl(int tid, double* __restrict__ COL, double* __restrict__ DINV, RowState<D>* st) {
  const Team<TEAM> t{tid};
  (void)t;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    double* col = COL + (k & 1) * D;
    ODEF_ROWS_PHASE(
      if (r < D && r >= k) {
        col[r] = L.lr[k];
        if (r == k) DINV[k] = (L.lr[k] > 0.0) ? 1.0 / L.lr[k] : 0.0;  // semi-definite rule: zero column
      }
    )
    ODEF_ROWS_PHASE(
      if (r < D && r > k) {
        const double lik = L.lr[k] * DINV[k];
_Pragma("unroll")
        for (int j = k + 1; j < D; ++j)
          if (j <= r) L.lr[j] -= lik * col[j];
        L.lr[k] = lik;
      }
    )
  }
}

// ---- the two halves of one step on the team's rows (shared by the smoother loop and the dense output, dense_rows.h).
// In (per lane r < D): L.xr = row r of P S P, L.mf = P m, L.pj / L.pij, L.csr = row r of S^s_+ (un-preconditioned), L.ms = m^s_+.
// rows_predict_phase:  L.yr / YL = row of Y = S A', L.mpred = (A m~)_r, L.lr = row r of B = A S A' + sigma2 Q (preconditioned),
//                      MM = row of M = P S^s_+ P - B, VDL = P m^s_+ - m^-
// rows_gain_phase:     B = L D L', G = Y B^-1, L.ms <- P^-1 (m~ + G delta), L.csr <- full row r of P^-1 (S + G M G') P^-1
template <int d, int q, int TEAM>
__device__ inline void rows_predict_phase(const PriorConsts& pc, const double (&pjv)[q + 1], double sigma2, int tid, double* __restrict__ ws,
                                          RowState<d*(q + 1)>* st) {
  constexpr int NB = q + 1, D = d * NB;
  using W = RowsWs<d, NB>;
  constexpr int LD = W::LD;
  const Team<TEAM> t{tid};
  (void)t;
  double* YL = ws + W::YL;
  double* MM = ws + W::MM;
  double* GG = ws + W::GG;
  double* COL = ws + W::COL;
  double* DINV = ws + W::DINV;
  double* VMT = ws + W::VMT;
  double* VDL = ws + W::VDL;
  (void)YL; (void)MM; (void)GG; (void)COL; (void)DINV; (void)VMT; (void)VDL;
  ODEF_ROWS_PHASE(
    if (r < D) {
_Pragma("unroll")
      for (int K = 0; K < NB; ++K)
_Pragma("unroll")
        for (int b = 0; b < d; ++b) {
          double acc = L.xr[K * d + b];
_Pragma("unroll")
          for (int k = K + 1; k < NB; ++k) acc += L.xr[k * d + b] * pc.At[K][k];
          L.yr[K * d + b] = acc;
          YL[r * LD + K * d + b] = acc;
        }
      VMT[r] = L.mf;
    }
  )
  // phase 2: B row, M row, m^- , delta ; publish M, delta
  ODEF_ROWS_PHASE(
    if (r < D) {
      const int J = r / d;
      const int a = r % d;
      double mp = L.mf;
_Pragma("unroll")
      for (int j = 0; j < NB; ++j)
        if (j > J) mp += L.atr[j] * VMT[j * d + a];
_Pragma("unroll")
      for (int c = 0; c < D; ++c) {
        double acc = L.yr[c];
_Pragma("unroll")
        for (int j = 0; j < NB; ++j)
          if (j > J) acc += L.atr[j] * YL[(j * d + a) * LD + c];
        if (a == c % d) acc += sigma2 * L.qtr[c / d];
        L.lr[c] = acc;
        MM[r * LD + c] = L.csr[c] * (L.pj * pjv[c / d]) - acc;  // M = P S^s_+ P - S^-
        ODEF_SCHED_FENCE();
      }
      L.mpred = mp;
      VDL[r] = L.pj * L.ms - mp;
    }
  )
}
template <int d, int q, int TEAM>
__device__ inline void rows_gain_phase(const double (&pijv)[q + 1], int tid, double* __restrict__ ws, RowState<d*(q + 1)>* st) {
  constexpr int NB = q + 1, D = d * NB;
  using W = RowsWs<d, NB>;
  constexpr int LD = W::LD;
  const Team<TEAM> t{tid};
  (void)t;
  double* YL = ws + W::YL;
  double* MM = ws + W::MM;
  double* GG = ws + W::GG;
  double* COL = ws + W::COL;
  double* DINV = ws + W::DINV;
  double* VMT = ws + W::VMT;
  double* VDL = ws + W::VDL;
  (void)YL; (void)MM; (void)GG; (void)COL; (void)DINV; (void)VMT; (void)VDL;
  rows_ldl<D, TEAM>(tid, COL, DINV, st);
  // publish the unit-lower factor (Y is no longer needed in LDS; every lane keeps its Y row in registers)
  ODEF_ROWS_PHASE(
    if (r < D) {
_Pragma("unroll")
      for (int c = 0; c < D; ++c)
        if (c < r) YL[r * LD + c] = L.lr[c];
    }
  )
  // G row: g L D L' = y  ->  forward with L' (unit), scale, backward with L (unit); publish G
  ODEF_ROWS_PHASE(
    if (r < D) {
_Pragma("unroll")
      for (int k = 0; k < D; ++k) {
        double acc = L.yr[k];
_Pragma("unroll")
        for (int c = 0; c < k; ++c) acc -= YL[k * LD + c] * L.yr[c];
        L.yr[k] = acc;
        ODEF_SCHED_FENCE();
      }
_Pragma("unroll")
      for (int k = 0; k < D; ++k) L.yr[k] *= DINV[k];
_Pragma("unroll")
      for (int k = D - 1; k >= 0; --k) {
        double acc = L.yr[k];
_Pragma("unroll")
        for (int c = k + 1; c < D; ++c) acc -= YL[c * LD + k] * L.yr[c];
        L.yr[k] = acc;
        ODEF_SCHED_FENCE();
      }
_Pragma("unroll")
      for (int c = 0; c < D; ++c) GG[r * LD + c] = L.yr[c];
    }
  )
  // mean, T = G M (row), S^s row = S + T G' ; publish the lower part for symmetrisation
  ODEF_ROWS_PHASE(
    if (r < D) {
      double acc = L.mf;
_Pragma("unroll")
      for (int k = 0; k < D; ++k) acc += L.yr[k] * VDL[k];
      L.ms = acc * L.pij;  // un-precondition (src/smoothing.jl:26)
      double tr[D];
_Pragma("unroll")
      for (int c = 0; c < D; ++c) tr[c] = 0.0;
_Pragma("unroll")
      for (int k = 0; k < D; ++k) {
_Pragma("unroll")
        for (int c = 0; c < D; ++c) tr[c] += L.yr[k] * MM[k * LD + c];
        ODEF_SCHED_FENCE();
      }
_Pragma("unroll")
      for (int c = 0; c < D; ++c) {
        if (c <= r) {
          double o = L.xr[c];
_Pragma("unroll")
          for (int k = 0; k < D; ++k) o += tr[k] * GG[c * LD + k];
          L.csr[c] = o * (L.pij * pijv[c / d]);
        }
        ODEF_SCHED_FENCE();
      }
    }
  )
  // the M buffer is free now: exchange the lower triangle so that every lane holds its full (symmetric) row
  ODEF_ROWS_PHASE(
    if (r < D) {
_Pragma("unroll")
      for (int c = 0; c < D; ++c)
        if (c <= r) MM[r * LD + c] = L.csr[c];
    }
  )
  ODEF_ROWS_PHASE(
    if (r < D) {
_Pragma("unroll")
      for (int c = 0; c < D; ++c)
        if (c > r) L.csr[c] = MM[c * LD + r];
    }
  )
}

// whole backward pass of trajectory i.  `st`: one RowState (device) / TEAM RowStates (host emulation).
template <int d, int q, int TEAM>
__device__ inline void smooth_rows_lane(const SmoothParams& P, long i, int tid, double* __restrict__ ws, RowState<d*(q + 1)>* st) {
  constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  using W = RowsWs<d, NB>;
  constexpr int LD = W::LD;
  static_assert(TEAM >= D, "row-per-lane smoother needs one lane per state component");
  const Team<TEAM> t{tid};
  (void)t;
  const long n = P.adaptive ? (long)P.nsaved[i] : P.n_save;
  const size_t N = (size_t)P.N;
  double* YL = ws + W::YL;
  double* MM = ws + W::MM;
  double* GG = ws + W::GG;
  double* COL = ws + W::COL;
  double* DINV = ws + W::DINV;
  double* VMT = ws + W::VMT;
  double* VDL = ws + W::VDL;
  const PriorConsts& pc = P.pc;

  // first and last record are copied (index 1 in Julia is never smoothed, src/smoothing.jl:11)
  ODEF_ROWS_PHASE(
    if (r < D) {
      for (int w = 0; w < 2; ++w) {
        const long s = w == 0 ? 0 : n - 1;
_Pragma("unroll")
        for (int J = 0; J < NB; ++J) {  // lane constants without dynamically indexed kernel-argument reads
          if (J == r / d) {
_Pragma("unroll")
            for (int j = 0; j < NB; ++j) {
              L.atr[j] = pc.At[J][j];
              L.qtr[j] = pc.Qt[J][j];
            }
          }
        }
        const double v = P.mean[((size_t)s * D + r) * N + i];
        P.smean[((size_t)s * D + r) * N + i] = v;
        L.ms = v;
_Pragma("unroll")
        for (int c = 0; c < D; ++c) {
          const double cv = P.cov[((size_t)s * TRI + symidx(r, c)) * N + i];
          if (c <= r) P.scov[((size_t)s * TRI + tri(r, c)) * N + i] = cv;
          L.csr[c] = cv;
        }
      }
    }
  )
  bool nan_seen = false;
  for (long s = n - 2; s >= 1; --s) {
    // preconditioner of this step (src/preconditioning.jl:1-17): table for fixed grids, per-trajectory otherwise
    double h;
    double pjv[NB], pijv[NB];
    if (P.adaptive) {
      h = P.tsave[(size_t)(s + 1) * N + i] - P.tsave[(size_t)s * N + i];
      double val = (h != 0.0) ? precond_val<q>(h) : 0.0;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        pjv[J] = val;
        pijv[J] = 1.0 / val;
        val *= h;
      }
    } else {
      h = P.hs[s];
      const double* __restrict__ tab = P.ptab + (size_t)P.tab_idx[s] * kTabStride;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        pjv[J] = tab[kTabPJ + J];
        pijv[J] = tab[kTabPIJ + J];
      }
    }
    if (h == 0.0) {  // src/smoothing.jl:13-16
      ODEF_ROWS_PHASE(
        if (r < D) {
          P.smean[((size_t)s * D + r) * N + i] = L.ms;
_Pragma("unroll")
          for (int c = 0; c < D; ++c)
            if (c <= r) P.scov[((size_t)s * TRI + tri(r, c)) * N + i] = L.csr[c];
        }
      )
      continue;
    }
    const double sigma2 = P.diff[(size_t)(s + 1) * N + i];
    // phase 1: load row r and precondition it
    ODEF_ROWS_PHASE(
      if (r < D) {
        double pj_r = pjv[0];
        double pij_r = pijv[0];
_Pragma("unroll")
        for (int J = 1; J < NB; ++J) {
          pj_r = (r / d == J) ? pjv[J] : pj_r;
          pij_r = (r / d == J) ? pijv[J] : pij_r;
        }
        L.pj = pj_r;
        L.pij = pij_r;
        L.mf = pj_r * P.mean[((size_t)s * D + r) * N + i];
_Pragma("unroll")
        for (int c = 0; c < D; ++c) L.xr[c] = P.cov[((size_t)s * TRI + symidx(r, c)) * N + i] * (pj_r * pjv[c / d]);
      }
    )
    rows_predict_phase<d, q, TEAM>(pc, pjv, sigma2, tid, ws, st);
    rows_gain_phase<d, q, TEAM>(pijv, tid, ws, st);
    ODEF_ROWS_PHASE(
      if (r < D) {
        nan_seen = nan_seen || !(L.ms == L.ms);
        P.smean[((size_t)s * D + r) * N + i] = L.ms;
_Pragma("unroll")
        for (int c = 0; c < D; ++c)
          if (c <= r) P.scov[((size_t)s * TRI + tri(r, c)) * N + i] = L.csr[c];
      }
    )
  }
  if (nan_seen) P.retcode[i] = 3;  // "NaNs after smoothing" (src/smoothing.jl:25)
}

}  // namespace odef
