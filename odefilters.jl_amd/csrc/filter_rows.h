// Fixed-step EK0/EK1 filter, row-per-lane teams: 16 lanes own one trajectory (D = d(q+1) <= 16), lane r keeps ROW r
// of the covariance (full symmetric row) and component r of the mean in registers; 4 trajectories per wavefront.
// Meant for SMALL ensembles: the lane-per-trajectory kernel (ek_lane.h) has N/64 wavefronts for 1 024 SIMDs, this one
// N/4.  MEASURED (MI355X, Lorenz-63 EK1(3), 1 024 steps, tools/rows_vs_lane.py): 5.48 / 5.62 / 5.93 / 7.96 ms at
// 1 024 / 2 048 / 4 096 / 8 192 trajectories against 5.80 ms for the lane kernel at every one of these sizes -- no
// gain worth a second code path: with one wavefront per SIMD either mapping is bound by the step's serial d x d
// chain (chol(W) -> sigma^2 -> six Cholesky columns -> three reflections -> y: ~27 dependent sqrt/div sequences),
// which every lane here repeats for itself.  The kernel is therefore OFF by default (kFilterRowsMaxN = 0,
// ek_kernels.h) and kept, tested, as the record of the experiment; ODEF_FILTER_ROWS_MAX_N=<n> selects it for
// ensembles below n.
// Same arithmetic as EKStep::run (ek_math.h; src/perform_step.jl:27-76), distributed:
//
//   P1  x~ = P x (own row), Y = X~ A' (own row, lane-local) -> LDS, m~ -> LDS
//   P2  m^- component, row of A Y (rows r+d, r+2d, ... read from LDS), m^- -> LDS
//   P3  every lane redundantly: f, J, z, H, W = H Q H', sigma^2 (d x d work); own row += sigma^2 Q
//       Cholesky of the first 2d columns, right-looking on FULL rows so that the trailing part stays the symmetric
//       Schur complement: column k is exchanged through LDS (double buffered, one team sync per column)
//   P4  L1 (D x 2d) -> LDS
//   P5  every lane redundantly: G = (H L1)' from the top 2d x 2d of L1, its Householder QR, y = R^-T z, loglik;
//       own row of L1 times Q (the reflectors), own mean component, Z row -> LDS
//   P6  lower part of the own row of Sigma_filt = Z Z' + Schur, un-precondition, store it (the record), publish it
//   P7  upper part of the own row from the other lanes (one symmetric matrix in all lanes)
//
// `sync` is a wave-scope fence (LDS operations of one wave complete in issue order), no s_barrier.  The body is written
// as PHASES like smooth_rows.h: the device runs every phase on its own FRow, the host emulation (tests/emul) runs each
// phase for all lanes in turn over an array of FRows -- same source, same arithmetic.
#pragma once
#include "ek_lane.h"
#include "team.h"

namespace odef {

constexpr int kFilterRowsTeam = 16;

template <int d, int NB>
struct FRowsWs {  // LDS workspace per team (doubles)
  static constexpr int D = d * NB, LD = team_ld(D), d2 = 2 * d;
  static constexpr int YL = 0;               // D x LD : Y = X~ A', later L1 (D x 2d)
  static constexpr int VM = YL + D * LD;     // D      : m~
  static constexpr int MP = VM + D;          // D      : m^-
  static constexpr int COL = MP + D;         // 2 x D  : column exchange of the Cholesky (double buffered)
  static constexpr int ZP = COL + 2 * D;     // D x d  : Z rows
  static constexpr int used = ZP + D * d;
  static constexpr int size = used + ((8 - used % 32) + 32) % 32;  // teams of a wave 16 banks apart
};

template <int d, int NB>
struct FRow {
  static constexpr int D = d * NB, d2 = 2 * d;
  double xr[D];        // row r of Sigma (un-preconditioned between steps), inside the step row of Sigma^- / Schur
  double yr[D];        // row r of Y
  double m;            // own mean component (un-preconditioned)
  double mt, mp;       // preconditioned / predicted
  double pj, pij;      // own preconditioner entry and its inverse
  double atr[MAXNB], qtr[MAXNB];  // row (r / d) of At and Qt
  // d x d quantities every lane computes for itself
  double z[d], H0[d][d], h1, Wdiag[d];
  double sigma2_pred, sigma2_local, sigma2_global, loglik_acc;
  int chol_fix;
};

#ifdef ODEF_HOST_EMUL
#define ODEF_FROWS_PHASE(...)                                  \
  for (int lane_ = 0; lane_ < TEAM; ++lane_) {                 \
    FRow<d, NB>& L = st[lane_];                                \
    const int r = lane_;                                       \
    (void)L; (void)r;                                          \
    __VA_ARGS__                                                \
  }
#else
#define ODEF_FROWS_PHASE(...)                                  \
  {                                                            \
    FRow<d, NB>& L = st[0];                                    \
    const int r = tid;                                         \
    (void)L; (void)r;                                          \
    __VA_ARGS__                                                \
  }                                                            \
  t.sync();
#endif

// whole time loop of trajectory i.  `st`: one FRow (device) / TEAM FRows (host emulation).
template <class RHS, int q, bool IS_EK1, bool EVERY, int TEAM>
__device__ inline void filter_rows_lane(const FilterParams& P, long i, int tid, double* __restrict__ ws, FRow<RHS::d, q + 1>* st) {
  constexpr int d = RHS::d, NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2, d2 = 2 * d, np = RHS::np;
  using W = FRowsWs<d, NB>;
  constexpr int LD = W::LD;
  static_assert(TEAM >= D, "row-per-lane filter needs one lane per state component");
  static_assert(LD >= d2, "L1 is stored in the Y buffer");
  const Team<TEAM> t{tid};
  (void)t;
  const size_t N = (size_t)P.N;
  double* YL = ws + W::YL;
  double* VM = ws + W::VM;
  double* MP = ws + W::MP;
  double* COL = ws + W::COL;
  double* ZP = ws + W::ZP;
  const PriorConsts& pc = P.pc;

  double pl[np > 0 ? np : 1];
#pragma unroll
  for (int k = 0; k < np; ++k) pl[k] = P.p_shared ? P.p[k] : P.p[(size_t)k * N + i];

  // initial state (src/state_initialization.jl:2-53): every lane runs the Taylor recursion and keeps its component
  ODEF_FROWS_PHASE(
    if (r < D) {
      double u0[d];
_Pragma("unroll")
      for (int a = 0; a < d; ++a) u0[a] = P.u0[(size_t)a * N + i];
      double m0[D];
      taylor_init<RHS, q>(u0, pl, m0);
      double mine = m0[0];
_Pragma("unroll")
      for (int k = 1; k < D; ++k) mine = (k == r) ? m0[k] : mine;
      L.m = mine;
_Pragma("unroll")
      for (int c = 0; c < D; ++c) L.xr[c] = 0.0;
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
      L.sigma2_global = 0.0;
      L.loglik_acc = 0.0;
      L.chol_fix = 0;
      if (EVERY) {
        P.mean[(size_t)r * N + i] = L.m;
_Pragma("unroll")
        for (int c = 0; c < D; ++c)
          if (c <= r) P.cov[(size_t)tri(r, c) * N + i] = 0.0;
        if (r == 0) P.diff[i] = 0.0;
      }
    }
  )

  for (long n = 0; n < P.nsteps; ++n) {
    const double* __restrict__ tab = P.ptab + (size_t)P.tab_idx[n] * kTabStride;  // wave-uniform
    double pjv[NB], pijv[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      pjv[J] = tab[kTabPJ + J];
      pijv[J] = tab[kTabPIJ + J];
    }
    const double pi0 = pijv[0], pi1 = pijv[1];
    // P1: x~ = P x (src/perform_step.jl:36-38), Y = X~ A' (own row), publish Y and m~
    ODEF_FROWS_PHASE(
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
        L.mt = pj_r * L.m;
_Pragma("unroll")
        for (int c = 0; c < D; ++c) L.xr[c] = L.xr[c] * (pj_r * pjv[c / d]);
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
        VM[r] = L.mt;
      }
    )
    // P2: m^- = A m~ (src/filtering.jl:22-25), row of A Y
    ODEF_FROWS_PHASE(
      if (r < D) {
        const int J = r / d;
        const int a = r % d;
        double mp = L.mt;
_Pragma("unroll")
        for (int j = 0; j < NB; ++j)
          if (j > J) mp += L.atr[j] * VM[j * d + a];
        L.mp = mp;
        MP[r] = mp;
_Pragma("unroll")
        for (int c = 0; c < D; ++c) {
          double acc = L.yr[c];
_Pragma("unroll")
          for (int j = 0; j < NB; ++j)
            if (j > J) acc += L.atr[j] * YL[(j * d + a) * LD + c];
          L.xr[c] = acc;
        }
      }
    )
    // P3: measure! (src/perform_step.jl:95-132), diffusion (src/diffusions.jl:72-80), own row of Sigma^-;
    //     publish the first Cholesky column
    ODEF_FROWS_PHASE(
      if (r < D) {
        double up[d];
        double du[d];
_Pragma("unroll")
        for (int a = 0; a < d; ++a) up[a] = pi0 * MP[a];
        RHS::f(up, pl, du);
_Pragma("unroll")
        for (int a = 0; a < d; ++a) L.z[a] = pi1 * MP[d + a] - du[a];
        if constexpr (IS_EK1) {
          double Jm[d][d];
          rhs_jacobian<RHS>(up, pl, Jm);
_Pragma("unroll")
          for (int rr = 0; rr < d; ++rr)
_Pragma("unroll")
            for (int a = 0; a < d; ++a) L.H0[rr][a] = (0.0 - Jm[rr][a]) * pi0;
        } else {
_Pragma("unroll")
          for (int rr = 0; rr < d; ++rr)
_Pragma("unroll")
            for (int a = 0; a < d; ++a) L.H0[rr][a] = 0.0;
        }
        const double h1 = pi1;
        L.h1 = h1;
        // M = H Q_L (d x 2d nonzero), W = M M' = H Q H'
        double Wm[d][d];
        {
          double M0[d][d];
          const double m1 = h1 * pc.QLt[1][1];
_Pragma("unroll")
          for (int rr = 0; rr < d; ++rr)
_Pragma("unroll")
            for (int a = 0; a < d; ++a) {
              double tt = (rr == a) ? h1 * pc.QLt[1][0] : 0.0;
              if constexpr (IS_EK1) tt += L.H0[rr][a] * pc.QLt[0][0];
              M0[rr][a] = tt;
            }
_Pragma("unroll")
          for (int rr = 0; rr < d; ++rr)
_Pragma("unroll")
            for (int s = 0; s <= rr; ++s) {
              double tt = (rr == s) ? m1 * m1 : 0.0;
_Pragma("unroll")
              for (int a = 0; a < d; ++a) tt += M0[rr][a] * M0[s][a];
              Wm[rr][s] = tt;
              Wm[s][rr] = tt;
            }
        }
_Pragma("unroll")
        for (int rr = 0; rr < d; ++rr) L.Wdiag[rr] = Wm[rr][rr];
        double sigma2_pred = 1.0;
        if (!P.fixed_diffusion) {
          double Lw[d][d];
          double Lwi[d];
          chol_small<d>(Wm, Lw, Lwi);
          double s = 0.0;
          double yw[d];
_Pragma("unroll")
          for (int rr = 0; rr < d; ++rr) {
            double tt = L.z[rr];
_Pragma("unroll")
            for (int c = 0; c < rr; ++c) tt -= Lw[rr][c] * yw[c];
            yw[rr] = tt * Lwi[rr];
            s += yw[rr] * yw[rr];
          }
          sigma2_pred = s / d;
          L.sigma2_local = sigma2_pred;
          L.sigma2_global = sigma2_pred;
        }
        L.sigma2_pred = sigma2_pred;
        // Sigma^- row: + sigma^2 Q (src/filtering.jl:34-35)
_Pragma("unroll")
        for (int c = 0; c < D; ++c)
          if (c % d == r % d) L.xr[c] += sigma2_pred * L.qtr[c / d];
        COL[r] = L.xr[0];
      }
    )
    // Cholesky of the first 2d columns (src/filtering.jl:36), right-looking; the trailing rows stay full and symmetric
#pragma unroll
    for (int k = 0; k < d2; ++k) {
      const double* col = COL + (k & 1) * D;
      double* ncol = COL + ((k + 1) & 1) * D;
      ODEF_FROWS_PHASE(
        if (r < D) {
          const double piv = col[k];
          const bool ok = piv > 0.0;  // a failing pivot is the reference's QR-fallback case (src/filtering.jl:38-47)
          const double lkk = ok ? sqrt(piv) : 0.0;
          const double inv = ok ? 1.0 / lkk : 0.0;
          if (r == 0) L.chol_fix += ok ? 0 : 1;
          if (r > k) {
            const double lrk = L.xr[k] * inv;
_Pragma("unroll")
            for (int j = k + 1; j < D; ++j) L.xr[j] -= lrk * (col[j] * inv);
            L.xr[k] = lrk;
          } else if (r == k) {
            L.xr[k] = lkk;
          }
          if (k + 1 < d2 && r >= k + 1) ncol[r] = L.xr[k + 1];
        }
      )
    }
    // P4: publish L1 (lower-trapezoidal D x 2d)
    ODEF_FROWS_PHASE(
      if (r < D) {
_Pragma("unroll")
        for (int c = 0; c < d2; ++c) YL[r * LD + c] = (c <= r) ? L.xr[c] : 0.0;
      }
    )
    // P5: G = (H L1)', Householder QR, y, log-likelihood; own row of L1 times Q; mean; Z row
    ODEF_FROWS_PHASE(
      if (r < D) {
        double G[d2][d];
_Pragma("unroll")
        for (int c = 0; c < d2; ++c)
_Pragma("unroll")
          for (int rr = 0; rr < d; ++rr) {
            double s = 0.0;
            if constexpr (IS_EK1) {
_Pragma("unroll")
              for (int k = c; k < d; ++k) s += L.H0[rr][k] * YL[k * LD + c];
            }
            if (d + rr >= c) s += L.h1 * YL[(d + rr) * LD + c];
            G[c][rr] = s;
          }
        double hv[d][d2];
        double hbeta[d];
        double R[d][d];
_Pragma("unroll")
        for (int k = 0; k < d; ++k) {
          double nrm2 = 0.0;
_Pragma("unroll")
          for (int ii = k; ii < d2; ++ii) nrm2 += G[ii][k] * G[ii][k];
          const double nrm = sqrt(nrm2);
          const double x0 = G[k][k];
          const double alpha = (x0 >= 0.0) ? -nrm : nrm;
          const double v0 = x0 - alpha;
          const double vtv = nrm2 - x0 * x0 + v0 * v0;
          const double beta = (vtv > 0.0) ? 2.0 / vtv : 0.0;
          hbeta[k] = beta;
          hv[k][k] = v0;
_Pragma("unroll")
          for (int ii = k + 1; ii < d2; ++ii) hv[k][ii] = G[ii][k];
          R[k][k] = alpha;
_Pragma("unroll")
          for (int c = k + 1; c < d; ++c) {
            double s = v0 * G[k][c];
_Pragma("unroll")
            for (int ii = k + 1; ii < d2; ++ii) s += hv[k][ii] * G[ii][c];
            s *= beta;
            G[k][c] -= s * v0;
_Pragma("unroll")
            for (int ii = k + 1; ii < d2; ++ii) G[ii][c] -= s * hv[k][ii];
            R[k][c] = G[k][c];
          }
        }
        double y[d];
        double zSz = 0.0;
        double detprod = 1.0;
        double logacc = 0.0;
_Pragma("unroll")
        for (int rr = 0; rr < d; ++rr) {
          double tt = L.z[rr];
_Pragma("unroll")
          for (int c = 0; c < rr; ++c) tt -= R[c][rr] * y[c];
          y[rr] = tt / R[rr][rr];
          zSz += y[rr] * y[rr];
          if constexpr (d <= 4) detprod *= R[rr][rr];
          else if (P.want_loglik) logacc += log(fabs(R[rr][rr]));
        }
        if (P.want_loglik) {
          if constexpr (d <= 4) logacc = log(fabs(detprod));
          L.loglik_acc += -0.5 * (zSz + 2.0 * logacc + d * 1.8378770664093453);  // src/perform_step.jl:66
        }
        if (P.fixed_diffusion) {  // src/diffusions.jl:11-36, 46-68
          const double diffusion_t = zSz / d;
          L.sigma2_local = diffusion_t;
          L.sigma2_global = static_diffusion_update<d>(P.fixed_diffusion, (int)n, L.sigma2_global, diffusion_t);
        }
        // update! (src/filtering.jl:79-91): own row of L1 times Q
        double w[d2];
_Pragma("unroll")
        for (int c = 0; c < d2; ++c) w[c] = (c <= r) ? L.xr[c] : 0.0;
_Pragma("unroll")
        for (int k = 0; k < d; ++k) {
          double s = 0.0;
_Pragma("unroll")
          for (int c = k; c < d2; ++c) s += w[c] * hv[k][c];
          s *= hbeta[k];
_Pragma("unroll")
          for (int c = k; c < d2; ++c) w[c] -= s * hv[k][c];
        }
        double tt = L.mp;
_Pragma("unroll")
        for (int rr = 0; rr < d; ++rr) tt -= w[rr] * y[rr];
        L.m = L.pij * tt;  // un-precondition (src/perform_step.jl:75)
_Pragma("unroll")
        for (int rr = 0; rr < d; ++rr) {
          ZP[r * d + rr] = w[d + rr];
          L.yr[rr] = w[d + rr];  // own Z row
        }
      }
    )
    // P6: Sigma_filt row = Z Z' + Schur, un-precondition (src/perform_step.jl:73-75); the lower part is the record
    ODEF_FROWS_PHASE(
      if (r < D) {
_Pragma("unroll")
        for (int c = 0; c < D; ++c) {
          if (c <= r) {
            double s = (c >= d2) ? L.xr[c] : 0.0;  // r >= c >= 2d: Schur complement
_Pragma("unroll")
            for (int rr = 0; rr < d; ++rr) s += L.yr[rr] * ZP[c * d + rr];
            s *= (L.pij * pijv[c / d]);
            L.xr[c] = s;
            YL[r * LD + c] = s;
          }
        }
        if (EVERY) {
          const size_t slot = (size_t)(n + 1);
          P.mean[(slot * D + r) * N + i] = L.m;
_Pragma("unroll")
          for (int c = 0; c < D; ++c)
            if (c <= r) P.cov[(slot * TRI + tri(r, c)) * N + i] = L.xr[c];
          if (r == 0) P.diff[slot * N + i] = L.sigma2_global;
        }
      }
    )
    // P7: the upper part of the own row from the other lanes' lower parts: every lane carries the SAME symmetric matrix
    // (A Y is symmetric only up to rounding; without this the two halves drift apart over the steps)
    ODEF_FROWS_PHASE(
      if (r < D) {
_Pragma("unroll")
        for (int c = 0; c < D; ++c)
          if (c > r) L.xr[c] = YL[c * LD + r];
      }
    )
  }
  bool finite = true;
  ODEF_FROWS_PHASE(
    if (r < D) {
      if (!EVERY) {
        P.mean[(size_t)r * N + i] = L.m;
_Pragma("unroll")
        for (int c = 0; c < D; ++c)
          if (c <= r) P.cov[(size_t)tri(r, c) * N + i] = L.xr[c];
        if (r == 0) P.diff[i] = L.sigma2_global;
      }
      VM[r] = (fabs(L.m) <= 1.79769313486231570815e+308) ? 0.0 : 1.0;
    }
  )
  ODEF_FROWS_PHASE(
    if (r == 0) {
      double bad = 0.0;
      for (int k = 0; k < D; ++k) bad += VM[k];
      finite = bad == 0.0;
      P.loglik[i] = L.loglik_acc;
      P.naccept[i] = (int)P.nsteps;
      P.nreject[i] = 0;
      P.nf[i] = (int)P.nsteps;
      P.njac[i] = IS_EK1 ? (int)P.nsteps : 0;
      P.nsaved[i] = EVERY ? (int)P.nsteps + 1 : 1;
      P.retcode[i] = finite ? 0 /*Success*/ : 3 /*Unstable*/;
    }
  )
}

}  // namespace odef
