// RTS smoother (smooth_all!, src/smoothing.jl:4-63) for small and sharded ensembles: 16 lanes per trajectory on the
// DPP broadcasts of team_vec.h, the backward-pass counterpart of rows_filter.h.  Lane r keeps row r of every D x D
// matrix (D <= 16); nothing but the two row exchanges of a step (rows r + d, r + 2d, .. of Y for A Y, and the final
// symmetrisation) goes through LDS, every "row K to everybody" / "column c to everybody" of the factorisation, the two
// substitutions and the two D x D products is ONE v_fmac_f64_dpp per multiply-add.
//
// Per step (src/smoothing.jl:31-63, src/filtering.jl:136-154), in preconditioned coordinates:
//   X = P Sigma_i P, m = P m_i                        own row / component, loaded symmetric from the packed record
//   Y = X A' (own row);  B = A Y + sigma^2 Q          predicted covariance Sigma^-_{i+1} (src/filtering.jl:33-48)
//   M = P Sigma^s_{i+1} P - B;  delta = P m^s_{i+1} - A m
//   B = L D L'                                        right-looking, row r in registers, no square roots
//   G = Y B^-1                                        own row: u L' = y, w = u D^-1, g L = w      (the gain, src/smoothing.jl:43)
//   m^s = m + G delta;  Sigma^s = X + G M G'          own row: T = G M, then T G'
// The textbook form X + G (Sigma^s_+ - Sigma^-) G' is the identity the reference's own test asserts for its stacked-QR
// Joseph form (test/filtering.jl:113); both agree to the oracle's rounding noise (tests/test_emul_parity.py).
//
// Memory: the loads of step i-1 are issued BEFORE the stores of step i, so that the wait for them (vmcnt counts loads
// and stores in one in-order queue) never includes a store round trip.
#pragma once
#include "rows_filter.h"
#include "rows_store.h"

namespace odef {

template <int d, int q, bool ADAPT>
struct RowsSmoother {
  static constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2, LD = tv::lds_ld(D);
  static constexpr int kExchange = tv::lds_rows(d, NB) * LD;            // the team's exchange rows (doubles)
  static constexpr int kLdsDoubles = kExchange + tv::kTeam * D;         // + the lanes' rows of Q (kept out of the registers)
  static_assert(D <= tv::kTeam, "row-per-lane smoother: one lane per state component");
  using TV = tv::TV;

  tv::TU off_mean, off_full[D], off_low[D], off_one;  // record offsets: full symmetric row (loads), lower part (stores)
  size_t N;

  __device__ inline void init_offsets(long N_, long i) {
    N = (size_t)N_;
    off_mean = tv::make_offsets(N_, i, [](int r) { return r < D ? (long)r : -1L; });
    static_for<0, D>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      off_full[c] = tv::make_offsets(N_, i, [](int r) { return r < D ? (long)symidx(r, c) : -1L; });
      off_low[c] = tv::make_offsets(N_, i, [](int r) { return (r < D && c <= r) ? (long)tri(r, c) : -1L; });
    });
    off_one = tv::make_offsets(N_, i, [](int r) { return r == 0 ? 0L : -1L; });
  }
  __device__ inline void load_record(const double* mean, const double* cov, long s, TV& m, TV (&x)[D]) const {
    const tv::Field fm(const_cast<double*>(mean) + (size_t)s * D * N, (size_t)D * N * sizeof(double));
    const tv::Field fc(const_cast<double*>(cov) + (size_t)s * TRI * N, (size_t)TRI * N * sizeof(double));
    m = tv::field_load(fm, off_mean);
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = tv::field_load(fc, off_full[c]);
  }
  __device__ inline void store_record(double* mean, double* cov, long s, const TV& m, const TV (&x)[D]) const {
    const tv::Field fm(mean + (size_t)s * D * N, (size_t)D * N * sizeof(double));
    const tv::Field fc(cov + (size_t)s * TRI * N, (size_t)TRI * N * sizeof(double));
    tv::field_store(fm, off_mean, m);
#pragma unroll
    for (int c = 0; c < D; ++c) tv::field_store(fc, off_low[c], x[c]);
  }

  // Whole backward pass of the team's trajectory; n_hi: largest record count among the workgroup's trajectories.
  // Workgroup-collective (rows_store.h): every team walks the slots n_hi - 1 .. 0; at its own last record a trajectory
  // starts (x^s = x_filt there, index 1 in Julia is never smoothed, src/smoothing.jl:11), before that it only takes
  // part in the stores' barrier.
  __device__ inline void run(const SmoothParams& P, const RowsTeam& tm, long n_hi) {
    const long i = tm.i;
    const tv::Lds lds{tm.lds_team};
    tv::lds_clear(lds, kExchange);
    init_offsets(P.N, i);
    RowsConsts<d, NB> lc;
    lc.init(P.pc);
    tv::lds_put_private<D>(lds, kExchange, lc.qm);  // row r of Q = Qt (x) I_d: 2 D registers the step can use otherwise
    tv::lds_sync();
    const PriorConsts& pc = P.pc;
    const long n = !tm.valid ? 0 : ADAPT ? (long)P.nsaved[i] : P.n_save;
    RowsSink<D, false, false> sink;
    sink.init(tm, P.N, kLdsDoubles, LD, P.smean, P.scov, nullptr, nullptr);

    TV ms = tv::splat(0.0), csr[D];
#pragma unroll
    for (int c = 0; c < D; ++c) csr[c] = tv::splat(0.0);
    RowsScale<d, NB> sc;
    int cur_tab = -1;
    bool nan_seen = false;
    // software pipeline: once a trajectory is under way, (mf, xr) already hold the filter record of the slot about to run
    TV mf, xr[D];
    bool have = false;
    for (long s = n_hi - 1; s >= 0; --s) {
      const bool mine = s <= n - 1;
      bool step = mine && s >= 1 && s <= n - 2;
      if (mine && !have) load_record(P.mean, P.cov, s, mf, xr);
      have = false;
      double h = 0.0;
      if (step) {
        if constexpr (ADAPT) {
          const tv::Field ft(const_cast<double*>(P.tsave) + (size_t)s * N, 2 * N * sizeof(double));  // slots s and s + 1
          h = tv::field_load_uniform(ft, N, i, 1) - tv::field_load_uniform(ft, N, i, 0);
        } else {
          h = uniform_load(P.hs + s);
        }
        // h == 0: a repeated save time (rejected attempt): the smoothed state carries over (src/smoothing.jl:13-16)
        step = h != 0.0;
      }
      if (mine && !step) {
        if (s == n - 1 || s == 0) {  // first and last record: the filter state itself
          ms = mf;
#pragma unroll
          for (int c = 0; c < D; ++c) csr[c] = xr[c];
          sink.stage_cov(lds, LD, csr);
        }
        have = s >= 1;
        if (have) load_record(P.mean, P.cov, s - 1, mf, xr);
      }
      if (step) {
      if constexpr (ADAPT) {
        double tab[kTabStride];
        precond_table_fast<q, NB>(h, tab);
        sc.set(LocalTab{tab});
      } else {
        const int ti = uniform_load(P.tab_idx + s);
        if (ti != cur_tab) {
          sc.set(GlobalTab{P.ptab + (size_t)ti * kTabStride});
          cur_tab = ti;
        }
      }
      double sigma2;
      {
        const tv::Field fd(const_cast<double*>(P.diff) + (size_t)(s + 1) * N, N * sizeof(double));
        sigma2 = tv::field_load_uniform(fd, N, i, 0);
      }
      // x~ = P x_i (src/smoothing.jl:23), Y = X~ A' (own row)
      const TV mt = sc.pj * mf;
      TV xs[D], yr[D];
#pragma unroll
      for (int c = 0; c < D; ++c) xs[c] = xr[c] * sc.f[c / d];
#pragma unroll
      for (int K = 0; K < NB; ++K)
#pragma unroll
        for (int b = 0; b < d; ++b) {
          TV acc = xs[K * d + b];
#pragma unroll
          for (int k = K + 1; k < NB; ++k) acc = tv::fma(xs[k * d + b], pc.At[K][k], acc);
          yr[K * d + b] = acc;
        }
      // the next slot's record: its loads go out before this slot's stores
      have = true;
      load_record(P.mean, P.cov, s - 1, mf, xr);
      // m^- = A m~ ; B = A Y + sigma^2 Q (row r from the rows r + d, r + 2d, .. of Y)
      TV mp = mt;
      static_for<1, NB>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        mp = tv::fma(lc.at[t], tv::shl<t * d>(mt), mp);
      });
      tv::lds_put_row<D>(lds, LD, yr);
      tv::lds_sync();
      TV lr[D];
#pragma unroll
      for (int c = 0; c < D; ++c) lr[c] = yr[c];
      static_for<1, NB>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        TV other[D];
        tv::lds_get_row<t * d, D>(lds, LD, other);
#pragma unroll
        for (int c = 0; c < D; ++c) lr[c] = tv::fma(lc.at[t], other[c], lr[c]);
      });
      tv::lds_sync();
      {
        TV qm[D];
        tv::lds_get_private<D>(lds, kExchange, qm);
#pragma unroll
        for (int c = 0; c < D; ++c) lr[c] = tv::fma(sigma2, qm[c], lr[c]);
      }
      // M = P Sigma^s_{i+1} P - Sigma^- ; delta = P m^s_{i+1} - m^-
      TV Mr[D];
#pragma unroll
      for (int c = 0; c < D; ++c) Mr[c] = csr[c] * sc.f[c / d] - lr[c];
      const TV delta = sc.pj * ms - mp;
      // B = L D L' (right-looking; lane r ends with row r of the unit-lower factor in lr[c < r]; a non-positive pivot
      // zeroes its column, the semi-definite rule of ek_math.h) ...
      // ... with the forward substitution of G = Y B^-1 riding along (own row y: u L' = y, w = u D^-1): at column k the
      // entry w_k = u_k / D_k is final and the later entries lose B[j][k] w_k -- the same broadcasts as the factorisation
      static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const double piv = tv::bcast<k>(lr[k]);
        const bool ok = piv > 0.0;
        const double dinv = ok ? rcp_pos(ok ? piv : 1.0) : 0.0;
        const TV colk = lr[k];  // lane j: B[j][k] of the current Schur complement
        const TV lik = colk * dinv;
        const TV wk = yr[k] * dinv;
        tv::fb_cols<true, k + 1, D - k - 1>(lr, colk, lik);  // lr[j] -= l_rk B[j][k], j > k
        tv::fb_cols<true, k + 1, D - k - 1>(yr, colk, wk);   // y[j]  -= w_k  B[j][k], j > k
        lr[k] = lik;
        yr[k] = wk;
      });
      // ... and backwards: g L = w
      static_for<1, D>([&](auto jc) {
        constexpr int k = D - 1 - decltype(jc)::value;  // k = D-2 .. 0
        tv::fb_lanes<true, k + 1, D - k - 1>(yr[k], lr[k], yr);  // L[c][k] = bcast<c>(lr[k]), c > k
      });
      // m^s = m + G delta (src/smoothing.jl:50), un-preconditioned (:26)
      {
        TV acc = mt;
        tv::fb_lanes<false, 0, D>(acc, delta, yr);
        ms = sc.pij * acc;
      }
      // T = G M (row), Sigma^s = X + T G' (row)
      TV tr[D];
#pragma unroll
      for (int c = 0; c < D; ++c) tr[c] = tv::splat(0.0);
      static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        tv::fb_rows<false, k, D>(tr, Mr, yr[k]);  // tr[c] += M[k][c] g_k
      });
      static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        tv::fb_cols<false, 0, D>(xs, yr[k], tr[k]);  // xs[c] += G[c][k] t_k
      });
#pragma unroll
      for (int c = 0; c < D; ++c) xs[c] = xs[c] * sc.g[c / d];
      // one symmetric matrix in all lanes (lower triangle is the truth), the record is its lower part
      tv::lds_put_row<D>(lds, LD, xs);
      tv::lds_sync();
      tv::lds_get_sym<D>(lds, LD, lc.sym, csr);
      tv::lds_sync();
      nan_seen = nan_seen || tv::any_nan(ms);
      }  // step
      sink.put(s, mine, false, ms, csr, 0.0, 0.0);
    }
    if (nan_seen && tm.valid) P.retcode[i] = 3;  // "NaNs after smoothing" (src/smoothing.jl:25)
  }
};

}  // namespace odef
