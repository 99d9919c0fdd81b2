// EK0/EK1 filter for SMALL and SHARDED ensembles: 16 lanes (one DPP row) per trajectory, lane r keeps row r of the
// covariance (full symmetric row) and component r of the mean in registers; 4 trajectories per wavefront (team_vec.h).
// The lane-per-trajectory kernel (ek_lane.h) has N/64 wavefronts for the chip's 1 024 SIMDs, this one N/4: a
// 4 096-trajectory ensemble (BASELINE config 2), the 8 192 trajectories a GPU gets when 65 536 are sharded over 8
// (config 3), and the 16 384 of config 5 fill the chip instead of 6-25 % of it.
//
// One step (src/perform_step.jl:27-76), state dimension D = d(q+1) <= 16, in preconditioned coordinates:
//   x~ = P x                                         own row / component                       (:36-38)
//   m^- = A m~                                       NB-1 row shifts of the mean               (src/filtering.jl:22-25)
//   f, J, z, H, W = H Q H', sigma^2                  d x d, team-uniform (every lane its copy) (:95-132, src/diffusions.jl:72-80)
//   Sigma^- = A X~ A' + sigma^2 Q                    Y = X~ A' lane-local, rows r+d, r+2d, .. of Y through the team's LDS
//                                                     (= the Gram matrix of [A L  sigma Q_L], src/filtering.jl:34-35)
//   C = Sigma^- H' (row-local), S = H C (d x d: 2d*d broadcasts), S = L_s D_s L_s', K = C S^-1 (row-local solves)
//   m = m^- - K z                                                                               (src/filtering.jl:85-88)
//   Sigma = (I - K H) Sigma^- (I - K H)'             two rank-d updates of the own row, each ONE v_fmac_f64_dpp per entry:
//                                                     T = Sigma^- - K C',  Sigma = T - (T H') K'
//   un-precondition, symmetrise through LDS (lower triangle is the truth), store the lower part.
//
// The update is the reference's `L <- (I - K H) L^-` (src/filtering.jl:89) written on the Gram matrix: the carried
// quantity is Sigma = L L' (= SquarerootMatrix.mat, src/squarerootmatrix.jl:16), and (I-KH) Sigma^- (I-KH)' is the Gram
// matrix of (I-KH) L^- whatever K is -- positive semi-definite by construction, like the factor form.  It differs from
// the lane kernel's arithmetic (partial Cholesky of Sigma^- + Householder QR, ek_math.h) only in rounding: both sit at
// the oracle's own rounding-noise level (tests/test_emul_parity.py, tools/joseph_experiment.py), and neither needs the
// D-dimensional factorisation, whose d+2d pivots were the serial chain of the first row-team kernel (round 1).
// The two d x d factorisations are L D L' (reciprocals only, no square roots).
#pragma once
#include "ek_lane.h"
#include "team_vec.h"
#include "rows_store.h"

namespace odef {

// S = L D L' for a small symmetric matrix (lower triangle referenced): unit lower L (strict part written), the pivots
// and their reciprocals.  A non-positive pivot zeroes its column (reciprocal 0), the semi-definite rule of ek_math.h
// (the reference's Cholesky-failure branch, src/filtering.jl:38-47).
template <int n>
__device__ inline void ldl_small(const double (&S)[n][n], double (&L)[n][n], double (&dd)[n], double (&dinv)[n], int& fixes) {
#pragma unroll
  for (int j = 0; j < n; ++j) {
    double v[n > 1 ? n : 1];
    double s = S[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) {
      v[k] = L[j][k] * dd[k];
      s -= L[j][k] * v[k];
    }
    const bool ok = s > 0.0;
    const double inv = ok ? rcp_pos(ok ? s : 1.0) : 0.0;
    fixes += ok ? 0 : 1;
    dd[j] = ok ? s : 0.0;
    dinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      double t = S[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i][k] * v[k];
      L[i][j] = t * inv;
    }
  }
}

// Lane constants of a team: what lane r = (J, a) = (r / d, r % d) multiplies with.
template <int d, int NB>
struct RowsConsts {
  static constexpr int D = d * NB;
  tv::TV at[NB];  // at[t] = At[J][J + t] (0 beyond the last block): coefficient of row r + t d in row r of A (.)
  tv::TV qm[D];   // qm[c] = Qt[J][c / d] if c % d == a, else 0: row r of Q = Qt (x) I_d
  tv::SymIdx<D> sym;
  __device__ inline void init(const PriorConsts& pc) {
    sym.init(tv::lds_ld(D));
    double tab[tv::kTeam];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
#pragma unroll
      for (int r = 0; r < tv::kTeam; ++r) tab[r] = (r < D && r / d + t < NB) ? pc.At[(r / d) % NB][(r / d + t) % NB] : 0.0;
      at[t] = tv::lane_table(tab, tv::kTeam, 0.0);
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
#pragma unroll
      for (int r = 0; r < tv::kTeam; ++r) tab[r] = (r < D && r % d == c % d) ? pc.Qt[(r / d) % NB][c / d] : 0.0;
      qm[c] = tv::lane_table(tab, tv::kTeam, 0.0);
    }
  }
};

// Preconditioner of one step as the lanes use it (src/preconditioning.jl:1-17).
template <int d, int NB>
struct RowsScale {
  double pjv[NB], pijv[NB];  // P block values and their reciprocals (team-uniform)
  tv::TV pj, pij;            // the lane's own entry (1 in the idle lanes)
  tv::TV f[NB], g[NB];       // f[K] = pj * pjv[K] (precondition row entries of block K), g[K] = pij * pijv[K]
  template <class Tab>
  __device__ inline void set(const Tab& tab) {  // tab: one precond_fill table
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      pjv[J] = tab[kTabPJ + J];
      pijv[J] = tab[kTabPIJ + J];
    }
    pj = tv::block_table<d, NB>(pjv, 1.0);
    pij = tv::block_table<d, NB>(pijv, 1.0);
#pragma unroll
    for (int K = 0; K < NB; ++K) {
      f[K] = pj * pjv[K];
      g[K] = pij * pijv[K];
    }
  }
};

template <class RHS, int q, bool IS_EK1>
struct RowsStep {
  static constexpr int d = RHS::d, NB = q + 1, D = d * NB, LD = tv::lds_ld(D);
  static constexpr int kLdsDoubles = tv::lds_rows(d, NB) * LD;  // the team's exchange rows
  // adaptive kernel: + the lanes' rows of Q and the state before the attempt (restored when it is rejected), both kept
  // out of the registers
  static constexpr int kLdsDoublesAdaptive = kLdsDoubles + 2 * tv::kTeam * D;
  static_assert(D <= tv::kTeam, "row-per-lane filter: one lane per state component");
  static_assert(NB >= 2, "order >= 1");
  using TV = tv::TV;

  // One attempted step, in place: (m, xr) = cache.x -> cache.x_filt (src/perform_step.jl:27-76).
  // err_scale[r] = sqrt(diag(H (sigma2_local Q) H')) (src/perform_step.jl:148-158).
  __device__ static inline void run(const PriorConsts& pc, const RowsConsts<d, NB>& lc, const RowsScale<d, NB>& sc,
                                    const double* __restrict__ pl, int fixed_diffusion, bool want_loglik, int success_iter,
                                    double prev_global, const tv::Lds& lds, TV& m, TV (&xr)[D], double (&err_scale)[d],
                                    StepAux& aux, int qm_lds_off = -1) {  // qm_lds_off >= 0: the lanes' rows of Q are in LDS there
    const double pi0 = sc.pijv[0], pi1 = sc.pijv[1];
    // x~ = P x (src/perform_step.jl:36-38)
    const TV mt = sc.pj * m;
    TV xs[D];
#pragma unroll
    for (int c = 0; c < D; ++c) xs[c] = xr[c] * sc.f[c / d];
    // m^- = A m~ (src/filtering.jl:22-25): component (J, a) picks up the components (J + t, a)
    TV mp = mt;
    static_for<1, NB>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      mp = tv::fma(lc.at[t], tv::shl<t * d>(mt), mp);
    });
    // measure! (src/perform_step.jl:95-132), team-uniform
    double up[d], e1[d], du[d], z[d];
    tv::bcast_lanes<0, d>(mp, up);
    tv::bcast_lanes<d, d>(mp, e1);
#pragma unroll
    for (int a = 0; a < d; ++a) up[a] = pi0 * up[a];
    RHS::f(up, pl, du);
#pragma unroll
    for (int a = 0; a < d; ++a) z[a] = pi1 * e1[a] - du[a];
    double H0[d][d];  // H = (E1 - J E0) P^-1 -> blocks H0 = -J pi0, H1 = pi1 I;  EK0: H0 = 0
    if constexpr (IS_EK1) {
      double Jm[d][d];
      rhs_jacobian<RHS>(up, pl, Jm);
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
    const double h1 = pi1;
    // W = H Q H' from M = H Q_L (src/diffusions.jl:78)
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
    double sigma2_pred = 1.0;  // diffusion used inside predict_cov!
    if (!fixed_diffusion) {
      // DynamicDiffusion (src/diffusions.jl:72-80): sigma^2 = z' W^-1 z / d with W = L D L'
      double Lw[d][d], wd[d], wdinv[d];
      int wfix = 0;
      ldl_small<d>(W, Lw, wd, wdinv, wfix);
      double s = 0.0, yw[d];
#pragma unroll
      for (int r = 0; r < d; ++r) {
        double t = z[r];
#pragma unroll
        for (int c = 0; c < r; ++c) t -= Lw[r][c] * yw[c];
        yw[r] = t;
        s += t * t * wdinv[r];
      }
      sigma2_pred = s * (1.0 / d);
      aux.sigma2_local = sigma2_pred;
      aux.sigma2_global = sigma2_pred;
    }
    // Sigma^- = A X~ A' + sigma^2 Q (src/filtering.jl:34-35): Y = X~ A' in the own row ...
    TV zr[D];
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int b = 0; b < d; ++b) {
        TV acc = xs[K * d + b];
#pragma unroll
        for (int k = K + 1; k < NB; ++k) acc = tv::fma(xs[k * d + b], pc.At[K][k], acc);
        zr[K * d + b] = acc;
      }
    // ... and row r of A Y from the rows r + d, r + 2d, ... of Y
    tv::lds_put_row<D>(lds, LD, zr);
    tv::lds_sync();
    static_for<1, NB>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      TV other[D];
      tv::lds_get_row<t * d, D>(lds, LD, other);
#pragma unroll
      for (int c = 0; c < D; ++c) zr[c] = tv::fma(lc.at[t], other[c], zr[c]);
    });
    tv::lds_sync();
    if (qm_lds_off >= 0) {
      TV qm[D];
      tv::lds_get_private<D>(lds, qm_lds_off, qm);
#pragma unroll
      for (int c = 0; c < D; ++c) zr[c] = tv::fma(sigma2_pred, qm[c], zr[c]);
    } else {
#pragma unroll
      for (int c = 0; c < D; ++c) zr[c] = tv::fma(sigma2_pred, lc.qm[c], zr[c]);
    }

    // C = Sigma^- H' (own row), S = H C = H Sigma^- H' (src/perform_step.jl:54)
    TV Cr[d];
#pragma unroll
    for (int a = 0; a < d; ++a) {
      TV t = h1 * zr[d + a];
      if constexpr (IS_EK1) {
#pragma unroll
        for (int k = 0; k < d; ++k) t = tv::fma(zr[k], H0[a][k], t);
      }
      Cr[a] = t;
    }
    double S[d][d];
    {
      double cb[2 * d][d];  // cb[k][b] = C[k][b]: the first 2d rows of C, to everybody
      static_for<0, 2 * d>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (IS_EK1 || k >= d) tv::bcast_vec<k, d>(Cr, cb[k]);
      });
#pragma unroll
      for (int a = 0; a < d; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          double s = h1 * cb[d + a][b];
          if constexpr (IS_EK1) {
#pragma unroll
            for (int k = 0; k < d; ++k) s += H0[a][k] * cb[k][b];
          }
          S[a][b] = s;
          S[b][a] = s;
        }
    }
    double Ls[d][d], sd[d], sdinv[d];
    ldl_small<d>(S, Ls, sd, sdinv, aux.chol_fix);
    // y = L_s^-1 z;  z' S^-1 z = sum y^2 / D_s;  log det S = log prod D_s
    double y[d], zSz = 0.0, detprod = 1.0, logacc = 0.0;
#pragma unroll
    for (int r = 0; r < d; ++r) {
      double t = z[r];
#pragma unroll
      for (int c = 0; c < r; ++c) t -= Ls[r][c] * y[c];
      y[r] = t;
      zSz += t * t * sdinv[r];
      if constexpr (d <= 4) detprod *= sd[r];
      else if (want_loglik) logacc += log(sd[r]);
    }
    if (want_loglik) {
      // logpdf(measurement, 0) (src/perform_step.jl:66); d <= 4: log det S = log prod D_k is left to the caller's LogDetAcc
      aux.loglik = -0.5 * (zSz + logacc + d * 1.8378770664093453);
    } else {
      aux.loglik = 0.0;
    }
    aux.det = (d <= 4) ? detprod : 1.0;
    if (fixed_diffusion) {  // FixedDiffusion / MAPFixedDiffusion (src/diffusions.jl:11-36, 46-68)
      const double diffusion_t = zSz * (1.0 / d);
      aux.sigma2_local = diffusion_t;
      aux.sigma2_global = static_diffusion_update<d>(fixed_diffusion, success_iter, prev_global, diffusion_t);
    }
#pragma unroll
    for (int r = 0; r < d; ++r) err_scale[r] = sqrt(aux.sigma2_local * W[r][r]);  // src/perform_step.jl:148-158

    // K = C S^-1 (src/filtering.jl:85-86), own row: forward with L_s', scale, backward with L_s
    TV Kr[d];
    {
      TV w[d];
#pragma unroll
      for (int a = 0; a < d; ++a) {
        TV t = Cr[a];
#pragma unroll
        for (int b = 0; b < a; ++b) t = tv::fma(w[b], -Ls[a][b], t);
        w[a] = t;
      }
#pragma unroll
      for (int a = 0; a < d; ++a) w[a] = w[a] * sdinv[a];
#pragma unroll
      for (int a = d - 1; a >= 0; --a) {
        TV t = w[a];
#pragma unroll
        for (int b = a + 1; b < d; ++b) t = tv::fma(Kr[b], -Ls[b][a], t);
        Kr[a] = t;
      }
    }
    // m = m^- + K (0 - z) (src/filtering.jl:88), un-preconditioned (src/perform_step.jl:75)
    {
      TV mf = mp;
#pragma unroll
      for (int a = 0; a < d; ++a) mf = tv::fma(Kr[a], -z[a], mf);
      m = sc.pij * mf;
    }
    // (I - K H) Sigma^- (I - K H)' (src/filtering.jl:89 on the Gram matrix): T = Sigma^- - K C' ...
#pragma unroll
    for (int a = 0; a < d; ++a) tv::fnma_bc_cols<0, D>(zr, Cr[a], Kr[a]);
    // ... U = T H' (zero in exact arithmetic: the part of the update a rounded K leaves behind), Sigma = T - U K'
    TV Ur[d];
#pragma unroll
    for (int a = 0; a < d; ++a) {
      TV t = h1 * zr[d + a];
      if constexpr (IS_EK1) {
#pragma unroll
        for (int k = 0; k < d; ++k) t = tv::fma(zr[k], H0[a][k], t);
      }
      Ur[a] = t;
    }
#pragma unroll
    for (int a = 0; a < d; ++a) tv::fnma_bc_cols<0, D>(zr, Kr[a], Ur[a]);
    // un-precondition (src/perform_step.jl:73-75); every lane ends with the SAME symmetric matrix: the lower triangle
    // is the truth (rows kept by different lanes are symmetric only up to rounding, and that antisymmetric part is
    // amplified by stiff problems -- round-1 finding on van der Pol, order 5)
#pragma unroll
    for (int c = 0; c < D; ++c) zr[c] = zr[c] * sc.g[c / d];
    tv::lds_put_row<D>(lds, LD, zr);
    tv::lds_sync();
    tv::lds_get_sym<D>(lds, LD, lc.sym, xr);
    tv::lds_sync();
  }
};

// Per-lane offsets of one record (layout of include/odefilter.h: MEAN [slot][D][N], COV_TRIL [slot][TRI][N],
// DIFFUSION / T [slot][N]): lane r owns mean component r and the lower part of covariance row r.
template <int D>
struct RowsRecord {
  static constexpr int TRI = D * (D + 1) / 2;
  tv::TU mean, cov[D], one;  // `one`: lane 0 only (the per-slot scalars)
  size_t N;
  __device__ inline void init(long N_, long i) {
    N = (size_t)N_;
    mean = tv::make_offsets(N_, i, [](int r) { return r < D ? (long)r : -1L; });
    static_for<0, D>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      cov[c] = tv::make_offsets(N_, i, [](int r) { return (r < D && c <= r) ? (long)tri(r, c) : -1L; });
    });
    one = tv::make_offsets(N_, i, [](int r) { return r == 0 ? 0L : -1L; });
  }
  __device__ inline void store(const FilterParams& P, long slot, const tv::TV& m, const tv::TV (&xr)[D], double diffusion) const {
    const tv::Field fm(P.mean + (size_t)slot * D * N, (size_t)D * N * sizeof(double));
    tv::field_store(fm, mean, m);
    const tv::Field fc(P.cov + (size_t)slot * TRI * N, (size_t)TRI * N * sizeof(double));
#pragma unroll
    for (int c = 0; c < D; ++c) tv::field_store(fc, cov[c], xr[c]);
    const tv::Field fd(P.diff + (size_t)slot * N, N * sizeof(double));
    tv::field_store_uniform(fd, one, diffusion);
  }
  __device__ inline void store_time(const FilterParams& P, long slot, double t) const {
    const tv::Field ft(P.tsave + (size_t)slot * N, N * sizeof(double));
    tv::field_store_uniform(ft, one, t);
  }
};

template <class RHS, int q>
__device__ inline void rows_initial_state(const FilterParams& P, long i, double (&pl)[RHS::np > 0 ? RHS::np : 1],
                                          double (&u0)[RHS::d], tv::TV& m, tv::TV (&xr)[RHS::d * (q + 1)]) {
  constexpr int d = RHS::d, D = d * (q + 1), np = RHS::np;
#pragma unroll
  for (int k = 0; k < np; ++k) pl[k] = P.p_shared ? P.p[k] : P.p[(size_t)k * P.N + i];
#pragma unroll
  for (int a = 0; a < d; ++a) u0[a] = P.u0[(size_t)a * P.N + i];
  double m0[D];
  taylor_init<RHS, q>(u0, pl, m0);  // src/state_initialization.jl:2-53, every lane for itself
  m = tv::lane_table(m0, D, 0.0);
#pragma unroll
  for (int c = 0; c < D; ++c) xr[c] = tv::splat(0.0);
}

// Whole fixed-grid time loop of the team's trajectory (OrdinaryDiffEq's solve! loop on the device).
// Workgroup-collective when EVERY (rows_store.h): all 16 teams of the workgroup run the same number of steps.
template <class RHS, int q, bool IS_EK1, bool EVERY>
__device__ inline void rows_filter_fixed(const FilterParams& P, const RowsTeam& tm) {
  using S = RowsStep<RHS, q, IS_EK1>;
  constexpr int d = S::d, NB = S::NB, D = S::D, np = RHS::np;
  const long i = tm.i;
  const tv::Lds lds{tm.lds_team};
  tv::lds_clear(lds, S::kLdsDoubles);
  double pl[np > 0 ? np : 1], u0[d];
  tv::TV m, xr[D];
  rows_initial_state<RHS, q>(P, i, pl, u0, m, xr);
  RowsConsts<d, NB> lc;
  lc.init(P.pc);
  RowsSink<D, true, false> sink;
  if constexpr (EVERY) {
    sink.init(tm, P.N, S::kLdsDoubles, S::LD, P.mean, P.cov, P.diff, nullptr);
    sink.stage_cov(lds, S::LD, xr);
    sink.put(0, tm.valid, false, m, xr, 0.0, 0.0);
  }

  RowsScale<d, NB> sc;
  int cur_tab = -1;
  double loglik = 0.0, gdiff = 0.0;
  LogDetAcc lda;
  lda.init();
  int chol_fix = 0;
  for (long n = 0; n < P.nsteps; ++n) {
    const int ti = uniform_load(P.tab_idx + n);  // wave-uniform, scalar load
    if (ti != cur_tab) {                          // the lanes' scale factors change only when the step size does
      sc.set(GlobalTab{P.ptab + (size_t)ti * kTabStride});
      cur_tab = ti;
    }
    double es[d];
    StepAux aux;
    aux.chol_fix = 0;
    S::run(P.pc, lc, sc, pl, P.fixed_diffusion, P.want_loglik != 0, (int)n, gdiff, lds, m, xr, es, aux);
    loglik += aux.loglik;
    if (P.want_loglik) lda.mul(aux.det);
    gdiff = aux.sigma2_global;
    chol_fix += aux.chol_fix;
    if constexpr (EVERY) sink.put(n + 1, tm.valid, false, m, xr, gdiff, 0.0);
  }
  if constexpr (!EVERY) {
    if (tm.valid) {  // one record per solve: stored directly (32-byte pieces, once)
      RowsRecord<D> rec;
      rec.init(P.N, i);
      rec.store(P, 0, m, xr, gdiff);
    }
  }
  (void)chol_fix;
  const bool bad = tv::team_any(tv::nonfinite_flag(m));
  if (tm.valid && tv::is_lane0()) {
    P.loglik[i] = P.want_loglik ? loglik - 0.5 * lda.log_value() : loglik;
    P.naccept[i] = (int)P.nsteps;
    P.nreject[i] = 0;
    P.nf[i] = (int)P.nsteps;
    P.njac[i] = IS_EK1 ? (int)P.nsteps : 0;
    P.nsaved[i] = EVERY ? (int)P.nsteps + 1 : 1;
    P.retcode[i] = bad ? 3 /*Unstable*/ : 0 /*Success*/;
  }
}

// Adaptive filter of the team's trajectory: perform_step! + error estimate (src/perform_step.jl:78-92) + OrdinaryDiffEq's
// PI controller (third-party; exponents src/alg_utils.jl:23-24), as filter_adaptive_lane (ek_lane.h) -- same record
// layout (one record per ATTEMPTED step, a rejected attempt repeats the old state at the old time), same quirks
// (commit on EEst < 1, integ.u overwritten on rejection, a rejected step leaves P^-1 (P x)).  The previous state stays
// in registers, so a rejection re-reads nothing.  The d x d controller algebra is replicated in all 16 lanes of a team,
// so it is kept short: reciprocals by rcp_pos, log(qold) carried from the previous attempt, q11 only when rejected.
// The loop is workgroup-uniform: it runs until no team of the workgroup wants another attempt (RowsSink::put).
template <class RHS, int q, bool IS_EK1>
__device__ inline void rows_filter_adaptive(const FilterParams& P, const RowsTeam& tm) {
  using S = RowsStep<RHS, q, IS_EK1>;
  constexpr int d = S::d, NB = S::NB, D = S::D, np = RHS::np;
  const long i = tm.i;
  const tv::Lds lds{tm.lds_team};
  tv::lds_clear(lds, S::kLdsDoubles);
  double pl[np > 0 ? np : 1], u0[d];
  tv::TV m, xr[D];
  rows_initial_state<RHS, q>(P, i, pl, u0, m, xr);
  RowsConsts<d, NB> lc;
  lc.init(P.pc);
  constexpr int kQmOff = S::kLdsDoubles, kOldOff = S::kLdsDoubles + tv::kTeam * D;
  tv::lds_put_private<D>(lds, kQmOff, lc.qm);
  tv::lds_sync();
  RowsSink<D, true, true> sink;
  sink.init(tm, P.N, S::kLdsDoublesAdaptive, S::LD, P.mean, P.cov, P.diff, P.tsave);
  sink.stage_cov(lds, S::LD, xr);

  double ucur[d];
#pragma unroll
  for (int a = 0; a < d; ++a) ucur[a] = u0[a];
  const Controller& ct = P.ctrl;
  double t = P.t0, h = P.dt0, qold = ct.qoldinit, log_qold = log(ct.qoldinit), log_eest = 0.0;
  double loglik = 0.0, gdiff = 0.0;
  LogDetAcc lda;
  lda.init();
  int naccept = 0, nreject = 0, nsaved = 1, ret = 0;
  const long max_attempts = 20 * P.max_save + 1000;
  long attempts = 0;
  RowsScale<d, NB> sc;
  bool active = tm.valid && t < P.t1;
  bool any = sink.put(0, tm.valid, active, m, xr, 0.0, P.t0);
  for (long slot = 1; any; ++slot) {
    bool store = false;
    if (active) {
      if (nsaved >= P.max_save || attempts >= max_attempts) {
        ret = 1;  // MaxIters
        active = false;
      }
    }
    if (active) {
      ++attempts;
      h = fmin(h, ct.dtmax);
      h = fmin(h, P.t1 - t);  // tstop clipping
      if (!(h > ct.dtmin)) {
        ret = 2;  // DtLessThanMin
        active = false;
      }
    }
    if (active) {
      // P(h) (src/preconditioning.jl:1-17): h^(-q-1/2) and the running products; reciprocals for inv(P)
      {
        double tab[kTabStride];
        precond_table_fast<q, NB>(h, tab);
        sc.set(LocalTab{tab});
      }
      const tv::TV m_old = m;
      tv::lds_put_private<D>(lds, kOldOff, xr);  // the state before the attempt: needed again only when it is rejected
      double es[d];
      StepAux aux;
      aux.chol_fix = 0;
      S::run(P.pc, lc, sc, pl, P.fixed_diffusion, P.want_loglik != 0, naccept, gdiff, lds, m, xr, es, aux, kQmOff);
      // DiffEqBase.calculate_residuals! + ODE_DEFAULT_NORM (src/perform_step.jl:78-84)
      double unew[d];
      tv::bcast_lanes<0, d>(m, unew);
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < d; ++r) {
        const double e = h * es[r] * rcp_pos(P.abstol + fmax(fabs(ucur[r]), fabs(unew[r])) * P.reltol);
        acc += e * e;
      }
      double EEst = sqrt(acc * (1.0 / d));
      if (!(EEst == EEst) || !(fabs(EEst) <= 1.79769313486231570815e+308)) EEst = INFINITY;
#pragma unroll
      for (int r = 0; r < d; ++r) ucur[r] = unew[r];  // integ.u .= u_filt (src/perform_step.jl:86), also when rejected
      // stepsize_controller! (PI): q = EEst^beta1 / qold^beta2 / gamma, clamped
      double qq;
      if (EEst == 0.0) {
        qq = 1.0 / ct.qmax;
      } else {
        log_eest = log(EEst);
        qq = exp(ct.beta1 * log_eest - ct.beta2 * log_qold);
        qq = fmax(1.0 / ct.qmax, fmin(1.0 / ct.qmin, qq / ct.gamma));
      }
      const bool accepted = EEst <= 1.0;  // OrdinaryDiffEq accepts on <=
      if (!(EEst < 1.0)) {
        // x_filt is not committed (src/perform_step.jl:89): cache.x stays P^-1 (P x) of the old state (:73)
        m = sc.pij * (sc.pj * m_old);
        tv::lds_get_private<D>(lds, kOldOff, xr);
        sink.stage_cov(lds, S::LD, xr);  // the exchange rows hold the rejected candidate
      }
      if (accepted) {
        if (EEst < 1.0) {
          loglik += aux.loglik;
          if (P.want_loglik) lda.mul(aux.det);
        }
        if (qq <= ct.qsteady_max && qq >= ct.qsteady_min) qq = 1.0;
        qold = fmax(EEst, ct.qoldinit);
        log_qold = (EEst > ct.qoldinit) ? log_eest : log(ct.qoldinit);
        double tn = t + h;
        if (fabs(tn - P.t1) < 100.0 * 2.220446049250313e-16 * fmax(fabs(tn), fabs(P.t1))) tn = P.t1;
        t = tn;
        gdiff = aux.sigma2_global;
        ++naccept;
        h = h / qq;
      } else {
        ++nreject;
        const double q11 = (EEst == 0.0) ? 1.0 : exp(ct.beta1 * log_eest);
        h = h / fmin(1.0 / ct.qmin, q11 / ct.gamma);
      }
      // accepted: the new state at the new time; rejected: the old state again at the old time
      store = true;
      ++nsaved;
      if (accepted && tv::team_any(tv::nonfinite_flag(m))) {
        ret = 3;
        active = false;
      }
      if (!(t < P.t1)) active = false;
    }
    any = sink.put(slot, store, active, m, xr, gdiff, t);
  }
  (void)qold;
  if (tm.valid && tv::is_lane0()) {
    P.loglik[i] = P.want_loglik ? loglik - 0.5 * lda.log_value() : loglik;
    P.naccept[i] = naccept;
    P.nreject[i] = nreject;
    P.nf[i] = naccept + nreject;
    P.njac[i] = IS_EK1 ? naccept + nreject : 0;
    P.nsaved[i] = nsaved;
    P.retcode[i] = ret;
  }
}

}  // namespace odef
