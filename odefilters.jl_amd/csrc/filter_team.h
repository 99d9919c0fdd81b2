// Fixed-step EK0/EK1 filter with one TEAM of threads per trajectory (workgroup-per-instance
// mapping for state dimensions that do not fit one lane's registers: Pleiades, D = 168).
// Same arithmetic as EKStep::run (ek_math.h) -- partial Cholesky of the predicted covariance,
// Householder QR of (H L1)', orthogonal update -- phrased as team-parallel phases over matrices
// in a per-trajectory workspace (global memory, L2/MALL resident; LDS staging is the next step,
// DESIGN.md 7).  The carried covariance X lives in the workspace as a dense symmetric matrix.
#pragma once
#include "ek_lane.h"
#include "team.h"

namespace odef {

// Workspace of one trajectory, split in two: the D x D-sized buffers live in global memory
// (`big`, per trajectory), everything small -- including the parts that single threads walk
// serially (RHS/Jacobian, triangular solves, Householder scalars) -- lives in LDS (`small`).
template <int d, int NB>
struct FilterWs {
  static constexpr int D = d * NB, LD = team_ld(D), d2 = 2 * d, LDd = team_ld(d);
  // ---- big (global) ----
  static constexpr int X = 0;                    // D x LD   carried covariance / predicted cov / factor + Schur
  static constexpr int Y = X + D * LD;           // D x LD   X A'
  static constexpr int WB = Y + D * LD;          // D x d2   rows of L1, then L1 Q
  static constexpr int ZT = WB + D * d2;         // d x LD   Zp transposed: ZT[r][l] = (L1 Q)[l][d + r]
  static constexpr int size = ZT + d * LD;
  // ---- small (LDS) ----
  static constexpr int G = 0;                    // d2 x d   (H L1)'
  static constexpr int HV = G + d2 * d;          // d x d2   Householder vectors
  static constexpr int R = HV + d * d2;          // d x d    R factor of S
  static constexpr int H0 = R + d * d;           // d x d    -J pi0
  static constexpr int WM = H0 + d * d;          // d x LDd  H Q H'  (then its Cholesky factor)
  static constexpr int M0 = WM + d * LDd;        // d x d
  static constexpr int MV = M0 + d * d;          // m[D]
  static constexpr int MT = MV + D;              // mt[D]
  static constexpr int MP = MT + D;              // mp[D]
  static constexpr int Z = MP + D;               // z[d]
  static constexpr int YV = Z + d;               // y[d]
  static constexpr int UP = YV + d;              // u_pred[d]
  static constexpr int DU = UP + d;              // du[d]
  static constexpr int BETA = DU + d;            // beta[d]
  static constexpr int SC = BETA + d;            // scalars: [0] sigma2 [1] zSz [2] logdet [3] loglik acc [4] global diffusion
  static constexpr int COL = SC + 8;             // D        pivot column / per-row scalars
  static constexpr int small_size = COL + D;
};

struct TeamFilterParams {
  FilterParams fp;
  double* ws;  // [N][FilterWs::size]
};

template <class RHS, int q, bool IS_EK1, int TEAM>
struct TeamFilter {
  static constexpr int d = RHS::d, NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2, d2 = 2 * d;
  using W = FilterWs<d, NB>;
  static constexpr int LD = W::LD;

  // One step (src/perform_step.jl:27-76).  ws->MV / X: current filter state, updated in place.
  __device__ static inline void step(const Team<TEAM>& t, const PriorConsts& pc, const double* __restrict__ p,
                                     const double* __restrict__ tab, int fixed_diffusion, int success_iter,
                                     double* __restrict__ ws, double* __restrict__ sm) {
    double* X = ws + W::X;
    double* Y = ws + W::Y;
    double* WB = ws + W::WB;
    double* ZT = ws + W::ZT;
    double* G = sm + W::G;
    double* HV = sm + W::HV;
    double* R = sm + W::R;
    double* H0 = sm + W::H0;
    double* WM = sm + W::WM;
    double* M0 = sm + W::M0;
    double* m = sm + W::MV;
    double* mt = sm + W::MT;
    double* mp = sm + W::MP;
    double* z = sm + W::Z;
    double* y = sm + W::YV;
    double* up = sm + W::UP;
    double* du = sm + W::DU;
    double* beta = sm + W::BETA;
    double* sc = sm + W::SC;
    double* col = sm + W::COL;
    const double pi0 = tab[kTabPIJ + 0], pi1 = tab[kTabPIJ + 1], h1 = pi1;

    // x~ = P x (src/perform_step.jl:36-38)
    ODEF_TEAM_FOR(i, D) mt[i] = tab[kTabPJ + i / d] * m[i];
    ODEF_TEAM_FOR(e, D * D) {
      const int r = e / D, c = e % D;
      X[r * LD + c] *= tab[kTabPP + (r / d) * MAXNB + (c / d)];
    }
    t.sync();
    // m^- = A m~ ; u_pred (src/filtering.jl:22-25, src/perform_step.jl:43)
    ODEF_TEAM_FOR(i, D) {
      const int J = i / d, a = i % d;
      double s = mt[i];
      for (int j = J + 1; j < NB; ++j) s += pc.At[J][j] * mt[j * d + a];
      mp[i] = s;
      if (i < d) up[i] = pi0 * s;
    }
    team_mul_At<d, NB, TEAM>(t, pc, X, Y);  // Y = X A' (independent of the mean)
    t.sync();
    // measure! (src/perform_step.jl:95-132): vector field and Jacobian by one thread
    if (t.tid == 0) {
      double u_[d], du_[d];
      for (int a = 0; a < d; ++a) u_[a] = up[a];
      RHS::f(u_, p, du_);
      for (int a = 0; a < d; ++a) du[a] = du_[a];
      if constexpr (IS_EK1) RHS::jac(u_, p, *reinterpret_cast<double (*)[d][d]>(H0));  // raw J, scaled below
    }
    t.sync();
    ODEF_TEAM_FOR(a, d) z[a] = pi1 * mp[d + a] - du[a];
    // H0 = -J pi0 ; M0 = H0 QL00 + I h1 QL10 (src/diffusions.jl:78)
    ODEF_TEAM_FOR(e, d * d) {
      const int r = e / d, a = e % d;
      double h0 = 0.0;
      if constexpr (IS_EK1) h0 = (0.0 - H0[e]) * pi0;
      H0[e] = h0;
      M0[e] = h0 * pc.QLt[0][0] + (r == a ? h1 * pc.QLt[1][0] : 0.0);
    }
    t.sync();
    {
      const double m1 = h1 * pc.QLt[1][1];
      ODEF_TEAM_FOR(e, d * d) {
        const int r = e / d, s_ = e % d;
        double acc = (r == s_) ? m1 * m1 : 0.0;
        for (int a = 0; a < d; ++a) acc += M0[r * d + a] * M0[s_ * d + a];
        WM[r * W::LDd + s_] = acc;
      }
    }
    t.sync();
    double sigma2_pred = 1.0;
    if (!fixed_diffusion) {
      // sigma^2 = |Lw^-1 z|^2 / d  (src/diffusions.jl:72-80)
      team_cholesky<d, TEAM>(t, WM, d);
      if (t.tid == 0) {
        double acc = 0.0;
        for (int r = 0; r < d; ++r) {
          double s = z[r];
          for (int c = 0; c < r; ++c) s -= WM[r * W::LDd + c] * y[c];
          y[r] = s / WM[r * W::LDd + r];
          acc += y[r] * y[r];
        }
        sc[0] = acc / d;
        sc[4] = acc / d;
      }
      t.sync();
      sigma2_pred = sc[0];
    }
    // predict_cov! (src/filtering.jl:33-41): X = A Y + sigma2 Q, Cholesky of the first 2d columns
    team_A_mul_plusQ<d, NB, TEAM>(t, pc, Y, sigma2_pred, X);
    t.sync();
    team_cholesky_coalesced<D, TEAM>(t, X, d2, col);
    // G = (H L1)' ; rows of L1 into WB
    ODEF_TEAM_FOR(e, d2 * d) {
      const int c = e / d, r = e % d;
      double s = 0.0;
      if constexpr (IS_EK1) {
        for (int k = c; k < d; ++k) s += H0[r * d + k] * X[k * LD + c];
      }
      if (d + r >= c) s += h1 * X[(d + r) * LD + c];
      G[c * d + r] = s;
    }
    ODEF_TEAM_FOR(e, D * d2) {
      const int l = e / d2, c = e % d2;
      WB[e] = (c <= l) ? X[l * LD + c] : 0.0;
    }
    t.sync();
    // Householder QR of G (2d x d): every thread derives the same scalars; one thread per trailing column
    for (int k = 0; k < d; ++k) {
      double nrm2 = 0.0;
      for (int i = k; i < d2; ++i) nrm2 += G[i * d + k] * G[i * d + k];
      const double nrm = sqrt(nrm2);
      const double x0 = G[k * d + k];
      const double alpha = (x0 >= 0.0) ? -nrm : nrm;
      const double v0 = x0 - alpha;
      const double vtv = nrm2 - x0 * x0 + v0 * v0;
      const double bt = (vtv > 0.0) ? 2.0 / vtv : 0.0;
      t.sync();  // all threads have read column k before it is touched
      ODEF_TEAM_FOR(i, d2) {
        if (i >= k) HV[k * d2 + i] = (i == k) ? v0 : G[i * d + k];
      }
      if (t.tid == 0) {
        beta[k] = bt;
        R[k * d + k] = alpha;
      }
      ODEF_TEAM_FOR(cc, d) {
        if (cc > k) {
          double s = v0 * G[k * d + cc];
          for (int i = k + 1; i < d2; ++i) s += G[i * d + k] * G[i * d + cc];
          s *= bt;
          const double gk = G[k * d + cc] - s * v0;
          G[k * d + cc] = gk;
          for (int i = k + 1; i < d2; ++i) G[i * d + cc] -= s * G[i * d + k];
          R[k * d + cc] = gk;
        }
      }
      t.sync();
    }
    // y = R^-T z ; z'S^-1 z ; log det S  (src/perform_step.jl:66)
    if (t.tid == 0) {
      double zSz = 0.0, logacc = 0.0;
      for (int r = 0; r < d; ++r) {
        double s = z[r];
        for (int c = 0; c < r; ++c) s -= R[c * d + r] * y[c];
        y[r] = s / R[r * d + r];
        zSz += y[r] * y[r];
        logacc += log(fabs(R[r * d + r]));
      }
      sc[1] = zSz;
      sc[3] += -0.5 * (zSz + 2.0 * logacc + d * 1.8378770664093453);
      if (fixed_diffusion) {  // src/diffusions.jl:11-36
        const double dt_ = zSz / d;
        sc[0] = dt_;
        sc[4] = static_diffusion_update<d>(fixed_diffusion, success_iter, sc[4], dt_);
      }
    }
    // rows of L1 times Q (src/filtering.jl:85-89): per reflector, one dot product per row, then a
    // coalesced rank-1 update of the D x 2d block
    for (int k = 0; k < d; ++k) {
      ODEF_TEAM_FOR(l, D) {
        const double* w = WB + l * d2;
        double s = 0.0;
        for (int c = k; c < d2; ++c) s += w[c] * HV[k * d2 + c];
        col[l] = s * beta[k];
      }
      t.sync();
      ODEF_TEAM_FOR(e, D * d2) {
        const int l = e / d2, c = e % d2;
        if (c >= k) WB[e] -= col[l] * HV[k * d2 + c];
      }
      t.sync();
    }
    // Zp transposed, so that the Gram matrix below reads contiguous memory along its fast index
    ODEF_TEAM_FOR(e, d * D) {
      const int r = e / D, l = e % D;
      ZT[r * LD + l] = WB[l * d2 + d + r];
    }
    t.sync();
    ODEF_TEAM_FOR(l, D) {
      double s = mp[l];
      for (int r = 0; r < d; ++r) s -= WB[l * d2 + r] * y[r];
      m[l] = tab[kTabPIJ + l / d] * s;  // un-precondition (src/perform_step.jl:75)
    }
    // Sigma_filt = Zp Zp' + Schur, un-preconditioned, both triangles
    ODEF_TEAM_FOR(e, D * D) {
      const int i = e / D, j = e % D;
      if (j <= i) {
        double s = (j >= d2) ? X[i * LD + j] : 0.0;
        for (int r = 0; r < d; ++r) s += ZT[r * LD + i] * ZT[r * LD + j];
        s *= tab[kTabPIPI + (i / d) * MAXNB + (j / d)];
        X[i * LD + j] = s;
        X[j * LD + i] = s;
      }
    }
    t.sync();
  }

  // whole fixed-step solve of trajectory i
  __device__ static inline void run(const TeamFilterParams& TP, long i, int tid, double* __restrict__ sm) {
    const FilterParams& P = TP.fp;
    const Team<TEAM> t{tid};
    double* ws = TP.ws + (size_t)i * W::size;
    double* X = ws + W::X;
    double* m = sm + W::MV;
    double* sc = sm + W::SC;
    const size_t N = (size_t)P.N;
    __attribute__((unused)) double pl_local[RHS::np > 0 ? RHS::np : 1];
    const double* pl = pl_local;
    for (int k = 0; k < RHS::np; ++k) pl_local[k] = P.p_shared ? P.p[k] : P.p[(size_t)k * N + i];
    if (tid == 0) {
      double u0[d], m0[D];
      for (int a = 0; a < d; ++a) u0[a] = P.u0[(size_t)a * N + i];
      taylor_init<RHS, q>(u0, pl, m0);
      for (int k = 0; k < D; ++k) m[k] = m0[k];
      for (int k = 0; k < 8; ++k) sc[k] = 0.0;
    }
    ODEF_TEAM_FOR(e, D * LD) X[e] = 0.0;
    t.sync();
    auto save = [&](long slot) {
      ODEF_TEAM_FOR(k, D) P.mean[((size_t)slot * D + k) * N + i] = m[k];
      ODEF_TEAM_FOR(e, D * D) {
        const int a = e / D, b = e % D;
        if (b <= a) P.cov[((size_t)slot * TRI + tri(a, b)) * N + i] = X[a * LD + b];
      }
      if (tid == 0) P.diff[(size_t)slot * N + i] = (slot == 0 && P.everystep) ? 0.0 : sc[4];
    };
    if (P.everystep) save(0);
    for (long n = 0; n < P.nsteps; ++n) {
      const double* tab = P.ptab + (size_t)P.tab_idx[n] * kTabStride;
      step(t, P.pc, pl, tab, P.fixed_diffusion, (int)n, ws, sm);
      if (P.everystep) save(n + 1);
    }
    if (!P.everystep) save(0);
    if (tid == 0) {
      P.loglik[i] = sc[3];
      P.naccept[i] = (int)P.nsteps;
      P.nreject[i] = 0;
      P.nf[i] = (int)P.nsteps;
      P.njac[i] = IS_EK1 ? (int)P.nsteps : 0;
      P.nsaved[i] = P.everystep ? (int)P.nsteps + 1 : 1;
      bool ok = true;
      for (int k = 0; k < D; ++k) ok = ok && (fabs(m[k]) <= 1.79769313486231570815e+308);
      P.retcode[i] = ok ? 0 : 3;
    }
  }
};

}  // namespace odef
