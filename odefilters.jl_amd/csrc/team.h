// Team-cooperative small dense algebra: TEAM threads work on ONE trajectory whose matrices
// live in a shared workspace (LDS for small D).  Every routine is a sequence of phases of the
// form "for (e = tid; e < n; e += TEAM) out[e] = f(inputs)" in which no thread reads what
// another thread writes in the same phase, separated by `sync()`.
//
//   TEAM == 1    host emulation / single thread: phases run sequentially, sync is a no-op
//   TEAM <= 64   sub-wave team (all teams of a wavefront run in lock step): sync is a
//                wave-scope memory fence + scheduling barrier (LDS operations of one wave are
//                performed in issue order, so no s_barrier is needed)
//   TEAM  > 64   whole workgroup: __syncthreads()
//
// Matrices are dense row-major D x D (symmetric ones stored in full) with an ODD leading dimension
// LD = D|1 so that the rows of a column access fall into distinct LDS banks.
#pragma once
#include "ek_math.h"

namespace odef {

template <int TEAM>
struct Team {
  int tid;
  __device__ inline void sync() const {
#ifndef ODEF_HOST_EMUL
    if constexpr (TEAM > 64) {
      __syncthreads();
    } else if constexpr (TEAM > 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#endif
  }
};

#define ODEF_TEAM_FOR(e, n) for (int e = t.tid; e < (n); e += TEAM)

__host__ __device__ constexpr int team_ld(int D) { return D | 1; }

// Y = X A'   (A = At (x) I_d block upper triangular):  Y[r][(K,b)] = sum_{k>=K} X[r][(k,b)] At[K][k]
template <int d, int NB, int TEAM>
__device__ inline void team_mul_At(const Team<TEAM>& t, const PriorConsts& pc, const double* __restrict__ X, double* __restrict__ Y) {
  constexpr int D = d * NB, LD = team_ld(D);
  ODEF_TEAM_FOR(e, D * D) {
    const int r = e / D, c = e % D, K = c / d, b = c % d;
    double s = X[r * LD + c];
    for (int k = K + 1; k < NB; ++k) s += X[r * LD + k * d + b] * pc.At[K][k];
    Y[r * LD + c] = s;
  }
}

// B = A Y + sigma2 Q   (full symmetric storage):  B[(J,a)][c] = sum_{j>=J} At[J][j] Y[(j,a)][c]
template <int d, int NB, int TEAM>
__device__ inline void team_A_mul_plusQ(const Team<TEAM>& t, const PriorConsts& pc, const double* __restrict__ Y, double sigma2,
                                        double* __restrict__ B) {
  constexpr int D = d * NB, LD = team_ld(D);
  ODEF_TEAM_FOR(e, D * D) {
    const int r = e / D, c = e % D, J = r / d, a = r % d;
    double s = Y[r * LD + c];
    for (int j = J + 1; j < NB; ++j) s += pc.At[J][j] * Y[(j * d + a) * LD + c];
    if (a == c % d) s += sigma2 * pc.Qt[J][c / d];
    B[r * LD + c] = s;
  }
}

// In-place right-looking Cholesky of the first `ncols` columns of the full-storage symmetric X
// (lower triangle is referenced and overwritten; the trailing block ends as the Schur complement).
// A non-positive pivot zeroes its column (semi-definite rule, see ek_math.h).
template <int D, int TEAM>
__device__ inline void team_cholesky(const Team<TEAM>& t, double* __restrict__ X, int ncols) {
  constexpr int LD = team_ld(D);
  for (int k = 0; k < ncols; ++k) {
    const double piv = X[k * LD + k];
    const bool ok = piv > 0.0;
    const double inv = ok ? 1.0 / piv : 0.0;
    // trailing update with the unscaled column: X[i][j] -= X[i][k] X[j][k] / piv   (k < j <= i)
    ODEF_TEAM_FOR(i, D) {
      if (i > k) {
        const double xik = X[i * LD + k] * inv;
        for (int j = k + 1; j <= i; ++j) X[i * LD + j] -= xik * X[j * LD + k];
      }
    }
    t.sync();
    const double rs = ok ? 1.0 / sqrt(piv) : 0.0;
    ODEF_TEAM_FOR(i, D) {
      if (i > k) X[i * LD + k] *= rs;
      else if (i == k) X[k * LD + k] = ok ? sqrt(piv) : 0.0;
    }
    t.sync();
  }
}

// Same factorisation for a whole-workgroup team on a matrix in global memory: every access is
// coalesced (threads run along rows), the pivot column is staged in the shared buffer `col` (D doubles)
// and the trailing update is applied to the full square so that both triangles stay valid.
template <int D, int TEAM>
__device__ inline void team_cholesky_coalesced(const Team<TEAM>& t, double* __restrict__ X, int ncols, double* __restrict__ col) {
  constexpr int LD = team_ld(D);
  constexpr int TX = TEAM >= 64 ? 64 : TEAM, TY = TEAM / TX;
  const int tx = t.tid % TX, ty = t.tid / TX;
  for (int k = 0; k < ncols; ++k) {
    ODEF_TEAM_FOR(i, D) col[i] = X[i * LD + k];
    t.sync();
    const double piv = col[k];
    const bool ok = piv > 0.0;
    const double inv = ok ? 1.0 / piv : 0.0;
    const double rs = ok ? 1.0 / sqrt(piv) : 0.0;
    for (int i = k + 1 + ty; i < D; i += TY) {
      const double ci = col[i] * inv;
      double* __restrict__ row = X + i * LD;
      for (int j = k + 1 + tx; j < D; j += TX) row[j] -= ci * col[j];
    }
    ODEF_TEAM_FOR(i, D) {
      if (i > k) X[i * LD + k] = col[i] * rs;
      else if (i == k) X[k * LD + k] = ok ? sqrt(piv) : 0.0;
    }
    t.sync();
  }
}

// Rows of G solve  G (L L') = Y  in place (row-parallel):  g = y L^-T, then g = g L^-1.
template <int D, int TEAM>
__device__ inline void team_solve_right_spd(const Team<TEAM>& t, const double* __restrict__ L, double* __restrict__ G) {
  constexpr int LD = team_ld(D);
  ODEF_TEAM_FOR(r, D) {
    double* g = G + r * LD;
    for (int k = 0; k < D; ++k) {
      double s = g[k];
      for (int c = 0; c < k; ++c) s -= L[k * LD + c] * g[c];
      const double lkk = L[k * LD + k];
      g[k] = (lkk != 0.0) ? s / lkk : 0.0;
    }
    for (int k = D - 1; k >= 0; --k) {
      double s = g[k];
      for (int c = k + 1; c < D; ++c) s -= L[c * LD + k] * g[c];
      const double lkk = L[k * LD + k];
      g[k] = (lkk != 0.0) ? s / lkk : 0.0;
    }
  }
}

// In-place transpose of a full-storage square matrix.
template <int D, int TEAM>
__device__ inline void team_transpose_inplace(const Team<TEAM>& t, double* __restrict__ X) {
  constexpr int LD = team_ld(D);
  ODEF_TEAM_FOR(e, D * D) {
    const int r = e / D, c = e % D;
    if (c < r) {
      const double a = X[r * LD + c], b = X[c * LD + r];
      X[r * LD + c] = b;
      X[c * LD + r] = a;
    }
  }
}

// Columns of GT solve (L L') GT = YT in place, one thread per COLUMN: thread r touches GT[k][r], so the threads of a
// wavefront read consecutive addresses (the row-parallel form above streams a private row per lane: 64 cache lines per
// load instruction).  Same arithmetic per right-hand side as team_solve_right_spd.
template <int D, int TEAM>
__device__ inline void team_solve_spd_columns(const Team<TEAM>& t, const double* __restrict__ L, double* __restrict__ GT) {
  constexpr int LD = team_ld(D);
  ODEF_TEAM_FOR(r, D) {
    for (int k = 0; k < D; ++k) {
      double s = GT[k * LD + r];
      for (int c = 0; c < k; ++c) s -= L[k * LD + c] * GT[c * LD + r];
      const double lkk = L[k * LD + k];
      GT[k * LD + r] = (lkk != 0.0) ? s / lkk : 0.0;
    }
    for (int k = D - 1; k >= 0; --k) {
      double s = GT[k * LD + r];
      for (int c = k + 1; c < D; ++c) s -= L[c * LD + k] * GT[c * LD + r];
      const double lkk = L[k * LD + k];
      GT[k * LD + r] = (lkk != 0.0) ? s / lkk : 0.0;
    }
  }
}

// ---- blocked factorisation and solves for the workgroup team on global-memory matrices ------------------------
// Block size kTB = 7 (divides D = 28 (q + 1)); every phase hands out whole kTB x kTB register tiles, so a thread
// issues ~0.6 global-memory instructions per FMA instead of 2-3 in the column-at-a-time forms above.
constexpr int kTB = 7;

// In-place blocked right-looking Cholesky of the full-storage symmetric X (lower triangle referenced and
// overwritten with L).  Per block column: diagonal block factored by one thread, panel rows solved against it
// (one thread per row), trailing tiles updated with the panel (one register tile per thread).  Zero-pivot rule as
// in team_cholesky.
template <int D, int TEAM>
__device__ inline void team_cholesky_blocked(const Team<TEAM>& t, double* __restrict__ X) {
  constexpr int LD = team_ld(D), NBK = D / kTB;
  static_assert(D % kTB == 0, "block size must divide the state dimension");
  for (int kb = 0; kb < NBK; ++kb) {
    const int k0 = kb * kTB;
    if (t.tid == 0) {
      double a[kTB][kTB];
#pragma unroll
      for (int r = 0; r < kTB; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) a[r][c] = X[(k0 + r) * LD + k0 + c];
#pragma unroll
      for (int k = 0; k < kTB; ++k) {
        const double piv = a[k][k];
        const bool ok = piv > 0.0;
        const double dg = ok ? sqrt(piv) : 0.0;
        const double rs = ok ? 1.0 / dg : 0.0;
        a[k][k] = dg;
#pragma unroll
        for (int r = k + 1; r < kTB; ++r) a[r][k] *= rs;
#pragma unroll
        for (int r = k + 1; r < kTB; ++r)
#pragma unroll
          for (int c = k + 1; c <= r; ++c) a[r][c] -= a[r][k] * a[c][k];
      }
#pragma unroll
      for (int r = 0; r < kTB; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) X[(k0 + r) * LD + k0 + c] = a[r][c];
    }
    t.sync();
    // panel: X[i][k0..k0+kTB) <- X[i][...] L_kk^-T for the rows below the block
    ODEF_TEAM_FOR(ii, D - k0 - kTB) {
      const int i = k0 + kTB + ii;
      double l[kTB][kTB], x[kTB];
#pragma unroll
      for (int r = 0; r < kTB; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) l[r][c] = X[(k0 + r) * LD + k0 + c];
#pragma unroll
      for (int c = 0; c < kTB; ++c) x[c] = X[i * LD + k0 + c];
#pragma unroll
      for (int c = 0; c < kTB; ++c) {
        double v = x[c];
#pragma unroll
        for (int j = 0; j < c; ++j) v -= x[j] * l[c][j];
        x[c] = (l[c][c] != 0.0) ? v / l[c][c] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < kTB; ++c) X[i * LD + k0 + c] = x[c];
    }
    t.sync();
    // trailing tiles (bi >= bj > kb): X_ij -= P_i P_j'
    const int nb = NBK - kb - 1;
    ODEF_TEAM_FOR(e, nb * (nb + 1) / 2) {
      int bi = 0;
      while ((bi + 1) * (bi + 2) / 2 <= e) ++bi;
      const int bj = e - bi * (bi + 1) / 2;
      const int i0 = (kb + 1 + bi) * kTB, j0 = (kb + 1 + bj) * kTB;
      double acc[kTB][kTB];
#pragma unroll
      for (int r = 0; r < kTB; ++r)
#pragma unroll
        for (int c = 0; c < kTB; ++c) acc[r][c] = X[(i0 + r) * LD + j0 + c];
#pragma unroll
      for (int k = 0; k < kTB; ++k) {
        double a[kTB], b[kTB];
#pragma unroll
        for (int r = 0; r < kTB; ++r) {
          a[r] = X[(i0 + r) * LD + k0 + k];
          b[r] = X[(j0 + r) * LD + k0 + k];
        }
#pragma unroll
        for (int r = 0; r < kTB; ++r)
#pragma unroll
          for (int c = 0; c < kTB; ++c) acc[r][c] -= a[r] * b[c];
      }
#pragma unroll
      for (int r = 0; r < kTB; ++r)
#pragma unroll
        for (int c = 0; c < kTB; ++c) X[(i0 + r) * LD + j0 + c] = acc[r][c];
    }
    t.sync();
  }
}

// (L L') GT = YT in place, blocked: GT[k][r] holds row k of the right-hand sides (column r = one system).
// Forward sweep with L, backward sweep with L'; per block row: the kTB x kTB triangular system of every column
// (one thread per column), then the remaining rows are updated with register tiles.
template <int D, int TEAM>
__device__ inline void team_solve_spd_blocked(const Team<TEAM>& t, const double* __restrict__ L, double* __restrict__ GT) {
  constexpr int LD = team_ld(D), NBK = D / kTB;
  static_assert(D % kTB == 0, "block size must divide the state dimension");
  for (int sweep = 0; sweep < 2; ++sweep) {
    for (int step = 0; step < NBK; ++step) {
      const int kb = sweep == 0 ? step : NBK - 1 - step;
      const int k0 = kb * kTB;
      ODEF_TEAM_FOR(r, D) {
        double l[kTB][kTB], z[kTB];
#pragma unroll
        for (int a = 0; a < kTB; ++a)
#pragma unroll
          for (int c = 0; c <= a; ++c) l[a][c] = L[(k0 + a) * LD + k0 + c];
#pragma unroll
        for (int a = 0; a < kTB; ++a) z[a] = GT[(k0 + a) * LD + r];
        if (sweep == 0) {
#pragma unroll
          for (int a = 0; a < kTB; ++a) {
            double v = z[a];
#pragma unroll
            for (int c = 0; c < a; ++c) v -= l[a][c] * z[c];
            z[a] = (l[a][a] != 0.0) ? v / l[a][a] : 0.0;
          }
        } else {
#pragma unroll
          for (int a = kTB - 1; a >= 0; --a) {
            double v = z[a];
#pragma unroll
            for (int c = a + 1; c < kTB; ++c) v -= l[c][a] * z[c];
            z[a] = (l[a][a] != 0.0) ? v / l[a][a] : 0.0;
          }
        }
#pragma unroll
        for (int a = 0; a < kTB; ++a) GT[(k0 + a) * LD + r] = z[a];
      }
      t.sync();
      // rows still to be solved: below the block (forward, factor L[i][k0+c]) / above it (backward, L[k0+c][i])
      const int nrem = sweep == 0 ? NBK - 1 - kb : kb;
      ODEF_TEAM_FOR(e, nrem * NBK) {
        const int bi = e / NBK, bc = e % NBK;
        const int i0 = (sweep == 0 ? kb + 1 + bi : bi) * kTB, r0 = bc * kTB;
        double acc[kTB][kTB];
#pragma unroll
        for (int a = 0; a < kTB; ++a)
#pragma unroll
          for (int c = 0; c < kTB; ++c) acc[a][c] = GT[(i0 + a) * LD + r0 + c];
#pragma unroll
        for (int k = 0; k < kTB; ++k) {
          double f[kTB], z[kTB];
#pragma unroll
          for (int a = 0; a < kTB; ++a) {
            f[a] = sweep == 0 ? L[(i0 + a) * LD + k0 + k] : L[(k0 + k) * LD + i0 + a];
            z[a] = GT[(k0 + k) * LD + r0 + a];
          }
#pragma unroll
          for (int a = 0; a < kTB; ++a)
#pragma unroll
            for (int c = 0; c < kTB; ++c) acc[a][c] -= f[a] * z[c];
        }
#pragma unroll
        for (int a = 0; a < kTB; ++a)
#pragma unroll
          for (int c = 0; c < kTB; ++c) GT[(i0 + a) * LD + r0 + c] = acc[a][c];
      }
      t.sync();
    }
  }
}

// C = op(A) B with a 7 x 7 (else 4 x 4) block of C per thread (full-storage D x D matrices in global memory):
// op(A) = A' when A_TRANSPOSED (A is given as its transpose).  Per k the thread loads 7 + 7 values for 49 FMAs;
// consecutive threads take consecutive column blocks, so the B loads of a wavefront are one contiguous 2 KB run and
// the A loads are wave-uniform.
template <int D, int TEAM, bool A_TRANSPOSED>
__device__ inline void team_gemm_blocked(const Team<TEAM>& t, const double* __restrict__ A, const double* __restrict__ B,
                                         double* __restrict__ C) {
  constexpr int LD = team_ld(D), BS = (D % kTB == 0) ? kTB : 4, NBK = D / BS;
  static_assert(D % BS == 0, "block size must divide the state dimension");
  ODEF_TEAM_FOR(blk, NBK * NBK) {
    const int r0 = (blk / NBK) * BS, c0 = (blk % NBK) * BS;
    double acc[BS][BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) acc[i][j] = 0.0;
    for (int k = 0; k < D; ++k) {
      double a[BS], b[BS];
#pragma unroll
      for (int i = 0; i < BS; ++i) a[i] = A_TRANSPOSED ? A[k * LD + r0 + i] : A[(r0 + i) * LD + k];
#pragma unroll
      for (int j = 0; j < BS; ++j) b[j] = B[k * LD + c0 + j];
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) acc[i][j] += a[i] * b[j];
    }
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) C[(r0 + i) * LD + c0 + j] = acc[i][j];
  }
}

// One Rauch-Tung-Striebel step for one trajectory (src/smoothing.jl:31-63, src/filtering.jl:136-154).
// Workspace `ws` (SmoothWs::size doubles): X | Y/G | M | vectors.  On entry X holds the filter covariance
// of time i (full symmetric, un-preconditioned), `mf` its mean, M / `ms` the smoothed covariance /
// mean of time i+1 (un-preconditioned).  On exit `ms` holds the smoothed mean of time i
// (un-preconditioned) and M holds G (Sigma^s_+ - Sigma^-) G' in preconditioned coordinates: the caller
// adds P Sigma P and un-preconditions (smooth_team_lane).
//   G = Sigma A' (Sigma^-)^-1 by two triangular solves against chol(Sigma^-) (the reference inverts
//   Sigma^- densely, src/squarerootmatrix.jl:42);
//   Sigma^s = Sigma + G (Sigma^s_+ - Sigma^-) G'  -- the identity test/filtering.jl:113 asserts for the
//   reference's stacked-QR Joseph form (src/smoothing.jl:53-57); both forms agree to the oracle's
//   own rounding noise on every test problem.
template <int d, int NB>
struct SmoothWs {
  static constexpr int D = d * NB, LD = team_ld(D), MAT = D * LD;
  static constexpr int X = 0, Y = MAT, M = 2 * MAT, MF = 3 * MAT, MS = MF + D, MP = MS + D, DL = MP + D, COL = DL + D, used = COL + D;
  // per-team stride: == 8 (mod 32) doubles, so the 4 teams of a wavefront start 16 banks apart
  static constexpr int size = used + ((8 - used % 32) + 32) % 32;
};

template <int d, int NB, int TEAM>
__device__ inline void team_smooth_step(const Team<TEAM>& t, const PriorConsts& pc, const double* __restrict__ tab, double sigma2,
                                        double* __restrict__ ws) {
  using W = SmoothWs<d, NB>;
  constexpr int D = W::D, LD = W::LD;
  double* X = ws + W::X;
  double* Y = ws + W::Y;
  double* M = ws + W::M;
  double* mf = ws + W::MF;
  double* ms = ws + W::MS;
  double* mp = ws + W::MP;
  double* dl = ws + W::DL;
  // precondition x_i and x_{i+1}^s  (src/smoothing.jl:23-24)
  ODEF_TEAM_FOR(e, D * D) {
    const int r = e / D, c = e % D;
    const double pp = tab[kTabPP + (r / d) * MAXNB + (c / d)];
    X[r * LD + c] *= pp;
    M[r * LD + c] *= pp;
  }
  ODEF_TEAM_FOR(i, D) {
    mf[i] *= tab[kTabPJ + i / d];
    ms[i] *= tab[kTabPJ + i / d];
  }
  t.sync();
  // predict (src/smoothing.jl:38): m^- = A m ; Y = Sigma A' ; B = A Y + sigma2 Q
  ODEF_TEAM_FOR(i, D) {
    const int J = i / d, a = i % d;
    double s = mf[i];
    for (int j = J + 1; j < NB; ++j) s += pc.At[J][j] * mf[j * d + a];
    mp[i] = s;
  }
  team_mul_At<d, NB, TEAM>(t, pc, X, Y);
  t.sync();
  // B = A Y + sigma2 Q (src/filtering.jl:34-35).  Only three D x D buffers are kept per trajectory, so
  // B overwrites X (Sigma); the caller re-reads Sigma from the filter record for the final sum.
  ODEF_TEAM_FOR(e, D * D) {
    const int r = e / D, c = e % D, J = r / d, a = r % d;
    double s = Y[r * LD + c];
    for (int j = J + 1; j < NB; ++j) s += pc.At[J][j] * Y[(j * d + a) * LD + c];
    if (a == c % d) s += sigma2 * pc.Qt[J][c / d];
    M[r * LD + c] -= s;  // M = Sigma^s_+ - Sigma^-
    X[r * LD + c] = s;   // X = Sigma^-  (Sigma itself is re-read by the caller)
  }
  ODEF_TEAM_FOR(i, D) dl[i] = ms[i] - mp[i];
  t.sync();
  if constexpr (D > 32) {  // (on the device: TEAM = 256; the host emulation runs the same branch with TEAM = 1)
    // Whole-workgroup team on matrices in global memory: everything below runs on G' (Y transposed in place), so
    // that consecutive threads always touch consecutive addresses.
    team_transpose_inplace<D, TEAM>(t, Y);
    if constexpr (D % kTB == 0) {
      team_cholesky_blocked<D, TEAM>(t, X);
      team_solve_spd_blocked<D, TEAM>(t, X, Y);  // Y = G'
    } else {
      team_cholesky_coalesced<D, TEAM>(t, X, D, ws + W::COL);
      team_solve_spd_columns<D, TEAM>(t, X, Y);  // Y = G'
    }
    t.sync();
    ODEF_TEAM_FOR(i, D) {
      double s = mf[i];
      for (int k = 0; k < D; ++k) s += Y[k * LD + i] * dl[k];
      ms[i] = s * tab[kTabPIJ + i / d];  // un-precondition (src/smoothing.jl:26)
    }
    team_gemm_blocked<D, TEAM, true>(t, Y, M, X);   // T = G M
    t.sync();
    team_gemm_blocked<D, TEAM, false>(t, X, Y, M);  // M <- T G'
    t.sync();
    return;
  }
  team_cholesky<D, TEAM>(t, X, D);
  // G = Y (Sigma^-)^-1, rows in place in Y  (src/smoothing.jl:42-43)
  team_solve_right_spd<D, TEAM>(t, X, Y);
  t.sync();
  // mean (src/smoothing.jl:44) and T = G M  (into X, whose factor is no longer needed)
  ODEF_TEAM_FOR(i, D) {
    double s = mf[i];
    for (int k = 0; k < D; ++k) s += Y[i * LD + k] * dl[k];
    ms[i] = s * tab[kTabPIJ + i / d];  // un-precondition (src/smoothing.jl:26)
  }
  ODEF_TEAM_FOR(e, D * D) {
    const int r = e / D, c = e % D;
    double s = 0.0;
    for (int k = 0; k < D; ++k) s += Y[r * LD + k] * M[k * LD + c];
    X[r * LD + c] = s;
  }
  t.sync();
  // M <- G M G' (the caller adds Sigma and un-preconditions)
  ODEF_TEAM_FOR(e, D * D) {
    const int r = e / D, c = e % D;
    double s = 0.0;
    for (int k = 0; k < D; ++k) s += X[r * LD + k] * Y[c * LD + k];
    M[r * LD + c] = s;
  }
  t.sync();
}

}  // namespace odef
