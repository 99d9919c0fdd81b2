// EK0/EK1 filter (fixed-step and adaptive) for large state dimension (Pleiades: d = 28, D = 168): one workgroup per
// trajectory, 320 tile threads + one helper wavefront, with the packed covariance DISTRIBUTED IN REGISTERS: tile
// thread t owns one 7 x 7 tile of the lower triangle (24 x 25 / 2 = 300 tiles at D = 168) for the whole solve; LDS
// (<= 160 KB) is only the exchange medium.  Same arithmetic as EKStep::run (ek_math.h):
//   congruence  A S A'                two stages through LDS, Y = S A' then A Y: <= 2 NB source tiles per thread
//   chol(H Q H'), sigma2              helper wavefront, in registers, concurrent with the congruence (wave_vec.h)
//   partial Cholesky (first 2d cols)  blocked: diagonal tile factored in registers, panel tiles solved, trailing
//                                     tiles updated with two panel tiles read from LDS
//   Householder QR of (H L1)', y      helper wavefront, row i of G in lane i, one shuffle butterfly per reflector
//   rows of L1 times Q                one thread per row, row in registers, reflectors broadcast from LDS
//   S_filt = Zp Zp' + Schur           own tile: 7 x 7 x d FMAs on 14 rows of Zp read from LDS; the Schur
//                                     part never leaves the registers
// The body is a sequence of PHASES separated by __syncthreads(); the host emulation (tests/emul) runs
// each phase for all threads in turn over an array of TileStates -- same source.
#pragma once
#include "ek_lane.h"
#include "team.h"
#include "wave_vec.h"

namespace odef {

constexpr int kTile = 7;
constexpr int kTilesThreads = 320;              // tile threads
constexpr int kTilesBlock = kTilesThreads + 64;  // + one helper wavefront (see ODEF_TILES_HELPER)

template <int d, int NB>
struct TilesLds {
  static constexpr int D = d * NB, nT = D / kTile, ntiles = nT * (nT + 1) / 2, d2 = 2 * d;
  static constexpr int LDL = d2 + 1, LDZ = d + 1, LDd = team_ld(d), T2 = kTile * kTile;
  // region A, time-shared: tile exchange EX[ntiles][49]  |  L1[D][LDL] + ZP[D][LDZ]
  static constexpr int EX = 0, L1 = 0, ZP = D * LDL;
  static constexpr int A_size = (ntiles * T2 > D * LDL + D * LDZ) ? ntiles * T2 : D * LDL + D * LDZ;
  // region B, time-shared: WM[d][LDd] + M0[d][d]  |  HV[d][d2] + R[d][d]
  static constexpr int WM = A_size, M0 = WM + d * LDd, HV = A_size, R = HV + d * d2;
  static constexpr int B_size = (d * LDd + d * d > d * d2 + d * d) ? d * LDd + d * d : d * d2 + d * d;
  static constexpr int G = A_size + B_size;  // d2 x d
  static constexpr int H0 = G + d2 * d;      // d x d
  static constexpr int MV = H0 + d * d, MT = MV + D, MP = MT + D, Z = MP + D, YV = Z + d, UP = YV + d, DU = UP + d;
  static constexpr int BETA = DU + d, SC = BETA + d, COL = SC + 8, COLEND = COL + D + d;  // col: D entries + d column sums
  // adaptive stepping: this attempt's preconditioner table, diag(H Q H') for the error estimate, integ.u, control words
  static constexpr int TAB = COLEND, WD = TAB + kTabStride, UC = WD + d, CTL = UC + d, size = CTL + 8;
  static_assert(d % kTile == 0, "tile size must divide the ODE dimension");
  static_assert(ntiles <= kTilesThreads && D <= kTilesThreads, "one thread per tile / per state row");
};

struct TileState {
  double x[kTile][kTile];
  double aj[MAXNB], ak[MAXNB];  // rows Jb = I/tpb and Kb = J/tpb of At: coefficients of this tile's congruence
  double qjk;                   // Qt[Jb][Kb]
  int I, J, tile;               // tile row / column / linear index; I < 0: this thread owns no tile
};

// In-kernel cycle stamps at phase boundaries: ONLY in the diagnostic build of tools/tiles_stamps.hip
// (-DODEF_TILES_STAMPS); the product kernel contains none.
#ifdef ODEF_TILES_STAMPS
#define ODEF_STAMP(k)                                                                        \
  if (tid_dev == 0 && blockIdx.x == 0 && g_stamp_buf) {                                                       \
    unsigned long long t_;                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    g_stamp_buf[(k)] += t_ - g_stamp_last;                                                   \
    g_stamp_last = t_;                                                                       \
  }
__device__ unsigned long long* g_stamp_buf = nullptr;
__device__ unsigned long long g_stamp_last = 0;
#else
#define ODEF_STAMP(k)
#endif

// Wave specialisation.  The workgroup has kTilesThreads tile threads plus ONE helper wavefront that owns no
// tile.  step() and run() are compiled twice (template parameter HELPER) from the same source:
//   ODEF_TILES_PHASE(body)   tile threads run body; everybody (helper included) meets at the closing barrier,
//                            so both instantiations execute the same number of barriers by construction;
//   ODEF_TILES_HELPER(body)  only the helper wavefront runs body (written in the one-value-per-lane form of
//                            wave_vec.h, no barrier inside); the tile threads walk on to the next barrier.
// The small sequential factorisations (Cholesky of W, Householder QR of G, the triangular solves) live in helper
// sections: ~150 live registers that never meet the 7 x 7 tile a tile thread carries, and the Cholesky of W
// overlaps with the congruence.  Host emulation: one instantiation, helper sections run in line.
#ifdef ODEF_HOST_EMUL
#define ODEF_TILES_PHASE(...)                                              \
  for (int tid = 0; tid < kTilesThreads; ++tid) {                          \
    TileState& S = st[tid];                                                \
    (void)S;                                                               \
    __VA_ARGS__                                                            \
  }
#define ODEF_TILES_HELPER(...) { __VA_ARGS__ }
#else
#define ODEF_TILES_PHASE(...)                                              \
  if constexpr (!HELPER) {                                                 \
    const int tid = tid_dev;                                               \
    TileState& S = st[0];                                                  \
    (void)S; (void)tid;                                                    \
    __VA_ARGS__                                                            \
  }                                                                        \
  __syncthreads();
#define ODEF_TILES_HELPER(...) if constexpr (HELPER) { __VA_ARGS__ }
#endif

// step() has two callers per instantiation (the fixed-step and the adaptive kernel).  Left to the inliner it became an
// out-of-line function: LDS pointers decayed to generic ones (FLAT instead of DS instructions), the tile registers
// were spilled around the call, and the fixed-step kernel ran 1.5x slower.  Everything here is force-inlined.
#ifdef ODEF_HOST_EMUL
#define ODEF_TILES_FN static inline
#else
#define ODEF_TILES_FN __device__ __attribute__((always_inline)) static inline
#endif

template <class RHS, int q, bool IS_EK1>
struct TilesFilter {
  static constexpr int d = RHS::d, NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2, d2 = 2 * d;
  using W = TilesLds<d, NB>;
  static constexpr int NT = kTilesThreads, TS = kTile, T2 = TS * TS, tpb = d / TS;  // tiles per derivative block

  template <bool HELPER>
  ODEF_TILES_FN void step(const PriorConsts& pc, const double* __restrict__ p, const double* __restrict__ tab,
                                     int fixed_diffusion, int success_iter, double* __restrict__ sm, TileState* st,
                                     int tid_dev) {
    (void)tid_dev;
    double* EX = sm + W::EX;
    double* L1 = sm + W::L1;
    double* ZP = sm + W::ZP;
    double* WM = sm + W::WM;
    double* M0 = sm + W::M0;
    double* HV = sm + W::HV;
    double* G = sm + W::G;
    double* H0 = sm + W::H0;
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

    ODEF_STAMP(0)
    // x~ = P x (src/perform_step.jl:36-38): mean in LDS, covariance tile in registers
    ODEF_TILES_PHASE(
      if (tid < D) mt[tid] = tab[kTabPJ + tid / d] * m[tid];
      if (S.I >= 0) {
        const double pp = tab[kTabPP + (S.I / tpb) * MAXNB + (S.J / tpb)];
        _Pragma("unroll")
        for (int r = 0; r < TS; ++r)
          _Pragma("unroll")
          for (int c = 0; c < TS; ++c) {
            S.x[r][c] *= pp;
            EX[S.tile * T2 + r * TS + c] = S.x[r][c];
          }
      }
    )
    // m^- = A m~ , u_pred  (src/filtering.jl:22-25, src/perform_step.jl:43)
    ODEF_TILES_PHASE(
      if (tid < D) {
        const int J = tid / d;
        const int a = tid % d;
        double s = mt[tid];
        for (int j = J + 1; j < NB; ++j) s += pc.At[J][j] * mt[j * d + a];
        mp[tid] = s;
        if (tid < d) up[tid] = pi0 * s;
      }
    )
    ODEF_STAMP(1)
    // measure! (src/perform_step.jl:95-132): vector field and Jacobian by one thread
    if constexpr (HasTeamEval<RHS>::value) {
      // many-thread evaluation; the pair buffer borrows region B (W / M0 are built afterwards)
      static_assert(RHS::team_scratch <= W::B_size, "pair buffer must fit region B");
      ODEF_TILES_PHASE(
        RHS::team_eval_pairs(tid, up, WM);
      )
      ODEF_TILES_PHASE(
        RHS::team_eval_assemble(tid, NT, up, WM, du, IS_EK1 ? H0 : nullptr);  // raw J into H0, scaled below
      )
    } else {
      ODEF_TILES_PHASE(
        if (tid == 0) {
          double u_[d];
          double du_[d];
          for (int a = 0; a < d; ++a) u_[a] = up[a];
          RHS::f(u_, p, du_);
          for (int a = 0; a < d; ++a) du[a] = du_[a];
          if constexpr (IS_EK1) RHS::jac(u_, p, *reinterpret_cast<double (*)[d][d]>(H0));  // raw J, scaled below
        }
      )
    }
    ODEF_STAMP(2)
    ODEF_TILES_PHASE(
      if (tid < d) z[tid] = pi1 * mp[d + tid] - du[tid];
      for (int e = tid; e < d * d; e += NT) {  // H0 = -J pi0 ; M0 = H0 QL00 + I h1 QL10 (src/diffusions.jl:78)
        const int r = e / d;
        const int a = e % d;
        double h0 = 0.0;
        if constexpr (IS_EK1) h0 = (0.0 - H0[e]) * pi0;
        H0[e] = h0;
        M0[e] = h0 * pc.QLt[0][0] + (r == a ? h1 * pc.QLt[1][0] : 0.0);
      }
    )
    ODEF_TILES_PHASE(
      const double m1 = h1 * pc.QLt[1][1];
      for (int e = tid; e < d * d; e += NT) {
        const int r = e / d;
        const int s_ = e % d;
        double acc = (r == s_) ? m1 * m1 : 0.0;
        for (int a = 0; a < d; ++a) acc += M0[r * d + a] * M0[s_ * d + a];
        WM[r * W::LDd + s_] = acc;
        if (r == s_) sm[W::WD + r] = acc;  // diag(H Q H') for the error estimate (src/perform_step.jl:148-158)
      }
    )
    ODEF_STAMP(3)
    // sigma^2 = z' W^-1 z / d (src/diffusions.jl:72-80): Cholesky of W and the forward substitution in the
    // registers of the helper wavefront, CONCURRENT with the congruence of the tile threads below; the value is
    // picked up after the congruence's closing barrier (first phase of the Cholesky).
    if (!fixed_diffusion) {
      ODEF_TILES_HELPER(
        const double acc = wv::chol_quadform<d>(wv::lds(WM), W::LDd, wv::lds(z));
        wv::store_uniform(wv::lds(sc + 0), acc / d);
        wv::store_uniform(wv::lds(sc + 4), acc / d);
      )
    }
    ODEF_STAMP(4)
    // predict_cov! (src/filtering.jl:33-41): own tile of A S A' (sigma2 Q is added below), in TWO stages through the
    // tile exchange: Y = S A' (own tile: <= NB source tiles of the own tile ROW), then A Y (<= NB tiles of the own tile
    // COLUMN, all of them below the diagonal: row tile j*tpb+si >= I >= J).  The slowest thread gathers 2 NB tiles
    // instead of NB^2 (12 instead of 36 at order 5) -- the phases last as long as their slowest thread.
    ODEF_TILES_PHASE(
      if (S.I >= 0) {
        const int Jb = S.I / tpb;
        const int Kb = S.J / tpb;
        const int sj = S.J % tpb;
        const int rt = S.I;
        double acc[TS][TS];
_Pragma("unroll")
        for (int r = 0; r < TS; ++r)
_Pragma("unroll")
          for (int c = 0; c < TS; ++c) acc[r][c] = 0.0;
        static_for<0, NB * NB>([&](auto jk) {  // compile-time (j, k): j = own block row, k = source block column
          constexpr int j = decltype(jk)::value / NB;
          constexpr int k = decltype(jk)::value % NB;
          if (j == Jb && k >= Kb) {
            const double coef = S.ak[k];
            const int ct = k * tpb + sj;
            // The source tile is stored as (rt, ct) when rt >= ct and transposed as (ct, rt) otherwise.  The two
            // cases are separate code paths so that every LDS read is base + compile-time offset.
            auto gather = [&](auto lower_c) {
              constexpr bool lower = decltype(lower_c)::value;
              const double* src = EX + (lower ? (rt * (rt + 1) / 2 + ct) : (ct * (ct + 1) / 2 + rt)) * T2;
_Pragma("unroll")
              for (int r = 0; r < TS; ++r) {
                double v[TS];
_Pragma("unroll")
                for (int c = 0; c < TS; ++c) v[c] = src[lower ? r * TS + c : c * TS + r];
_Pragma("unroll")
                for (int c = 0; c < TS; ++c) acc[r][c] += coef * v[c];
              }
            };
            if constexpr (j > k) {
              gather(std::true_type{});
            } else if constexpr (j < k) {
              gather(std::false_type{});
            } else {
              if (rt >= ct)
                gather(std::true_type{});
              else
                gather(std::false_type{});
            }
          }
        });
_Pragma("unroll")
        for (int r = 0; r < TS; ++r)
_Pragma("unroll")
          for (int c = 0; c < TS; ++c) S.x[r][c] = acc[r][c];
      }
    )
    ODEF_TILES_PHASE(  // every S tile has been read: the exchange now carries Y
      if (S.I >= 0) {
_Pragma("unroll")
        for (int r = 0; r < TS; ++r)
_Pragma("unroll")
          for (int c = 0; c < TS; ++c) EX[S.tile * T2 + r * TS + c] = S.x[r][c];
      }
    )
    ODEF_TILES_PHASE(
      if (S.I >= 0) {
        const int Jb = S.I / tpb;
        const int si = S.I % tpb;
        double acc[TS][TS];
_Pragma("unroll")
        for (int r = 0; r < TS; ++r)
_Pragma("unroll")
          for (int c = 0; c < TS; ++c) acc[r][c] = 0.0;
        static_for<0, NB>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if (j >= Jb) {
            const double coef = S.aj[j];
            const int rt = j * tpb + si;
            const double* src = EX + (rt * (rt + 1) / 2 + S.J) * T2;
_Pragma("unroll")
            for (int r = 0; r < TS; ++r) {
              double v[TS];
_Pragma("unroll")
              for (int c = 0; c < TS; ++c) v[c] = src[r * TS + c];
_Pragma("unroll")
              for (int c = 0; c < TS; ++c) acc[r][c] += coef * v[c];
            }
          }
        });
_Pragma("unroll")
        for (int r = 0; r < TS; ++r)
_Pragma("unroll")
          for (int c = 0; c < TS; ++c) S.x[r][c] = acc[r][c];
      }
    )
    ODEF_STAMP(5)
    // Blocked right-looking Cholesky of the first 2d columns (tile columns 0 .. 2d/TS-1); the tiles right of
    // them end as the Schur complement.  Per tile column: (1) the diagonal tile is factored in registers,
    // (2) the panel tiles below it are solved against it and published, (3) every trailing tile subtracts
    // the product of its two panel tiles.  A failing pivot zeroes its column, as in the unblocked form
    // (the reference's QR-fallback case, src/filtering.jl:38-47).
    double* DT = col;  // TS x TS diagonal factor followed by the TS reciprocal pivots
    for (int kt = 0; kt < d2 / TS; ++kt) {
      ODEF_TILES_PHASE(
        if (kt == 0 && S.I >= 0 && (S.I % tpb) == (S.J % tpb)) {  // + sigma2 Q: the helper's sigma2 is ready now
          const double sigma2_pred = fixed_diffusion ? 1.0 : sc[0];
_Pragma("unroll")
          for (int r = 0; r < TS; ++r) S.x[r][r] += sigma2_pred * S.qjk;
        }
        if (S.I == kt && S.J == kt) {
          double invd[TS];
          static_for<0, TS>([&](auto kcc) {
            constexpr int kc = decltype(kcc)::value;
            const double piv = S.x[kc][kc];
            const bool ok = piv > 0.0;
            double root;
            double rroot;
            sqrt_and_rsqrt(piv, root, rroot);  // 56 dependent pivots per step sit on the workgroup's critical path
            const double dg = ok ? root : 0.0;
            const double rs = ok ? rroot : 0.0;
            invd[kc] = rs;
            S.x[kc][kc] = dg;
_Pragma("unroll")
            for (int r = kc + 1; r < TS; ++r) S.x[r][kc] *= rs;
_Pragma("unroll")
            for (int r = kc + 1; r < TS; ++r)
_Pragma("unroll")
              for (int c = kc + 1; c <= r; ++c) S.x[r][c] -= S.x[r][kc] * S.x[c][kc];
          });
_Pragma("unroll")
          for (int r = 0; r < TS; ++r)
_Pragma("unroll")
            for (int c = 0; c < TS; ++c) DT[r * TS + c] = (c <= r) ? S.x[r][c] : 0.0;
_Pragma("unroll")
          for (int c = 0; c < TS; ++c) DT[T2 + c] = invd[c];
        }
      )
      ODEF_TILES_PHASE(
        if (S.I > kt && S.J == kt) {
          double l[TS][TS];
          double invd[TS];
_Pragma("unroll")
          for (int r = 0; r < TS; ++r)
_Pragma("unroll")
            for (int c = 0; c < TS; ++c) l[r][c] = DT[r * TS + c];
_Pragma("unroll")
          for (int c = 0; c < TS; ++c) invd[c] = DT[T2 + c];
          static_for<0, TS>([&](auto cc) {  // X <- X L^-T, column by column; the TS rows are independent
            constexpr int c = decltype(cc)::value;
_Pragma("unroll")
            for (int r = 0; r < TS; ++r) {
              double v = S.x[r][c];
_Pragma("unroll")
              for (int j = 0; j < c; ++j) v -= S.x[r][j] * l[c][j];
              S.x[r][c] = v * invd[c];
            }
          });
          double* dst = EX + S.I * T2;
_Pragma("unroll")
          for (int r = 0; r < TS; ++r)
_Pragma("unroll")
            for (int c = 0; c < TS; ++c) dst[r * TS + c] = S.x[r][c];
        }
      )
      ODEF_TILES_PHASE(
        if (S.I >= 0 && S.J > kt) {
          const double* pi_ = EX + S.I * T2;
          const double* pj_ = EX + S.J * T2;
          static_for<0, TS>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            double a[TS];
            double b[TS];
_Pragma("unroll")
            for (int r = 0; r < TS; ++r) {
              a[r] = pi_[r * TS + k];
              b[r] = pj_[r * TS + k];
            }
_Pragma("unroll")
            for (int r = 0; r < TS; ++r)
_Pragma("unroll")
              for (int c = 0; c < TS; ++c) S.x[r][c] -= a[r] * b[c];
          });
        }
      )
    }
    ODEF_STAMP(6)
    // publish L1 (lower-trapezoidal D x 2d) to LDS
    ODEF_TILES_PHASE(
      for (int e = tid; e < d2 * W::LDL; e += NT) L1[e] = 0.0;  // rows < 2d: the part above the diagonal
    )
    ODEF_TILES_PHASE(
      if (S.I >= 0 && S.J < d2 / TS) {
        _Pragma("unroll")
        for (int r = 0; r < TS; ++r)
          _Pragma("unroll")
          for (int c = 0; c < TS; ++c) {
            const int i = S.I * TS + r;
            const int j = S.J * TS + c;
            if (j <= i) L1[i * W::LDL + j] = S.x[r][c];
          }
      }
    )
    // G = (H L1)'  (2d x d)
    ODEF_TILES_PHASE(
      for (int e = tid; e < d2 * d; e += NT) {
        const int c = e / d;
        const int r = e % d;
        double s = 0.0;
        if constexpr (IS_EK1) {
          for (int k = c; k < d; ++k) s += H0[r * d + k] * L1[k * W::LDL + c];
        }
        if (d + r >= c) s += h1 * L1[(d + r) * W::LDL + c];
        G[c * d + r] = s;
      }
    )
    ODEF_STAMP(7)
    // Householder QR of G, y = R^-T z, z'S^-1 z and log det S (src/perform_step.jl:66) in the registers of the
    // helper wavefront (wave_vec.h); the reflectors, beta and y land in LDS, R never leaves the registers.
    ODEF_TILES_HELPER(
      double zSz;
      double logacc;
      wv::householder_qr_solve<d>(wv::lds(G), wv::lds(z), wv::lds(HV), wv::lds(beta), wv::lds(y), zSz, logacc);
      wv::store_uniform(wv::lds(sc + 1), zSz);
      wv::store_uniform(wv::lds(sc + 3), wv::load_uniform(wv::lds(sc + 3)) - 0.5 * (zSz + 2.0 * logacc + d * 1.8378770664093453));
      if (fixed_diffusion) {  // src/diffusions.jl:11-36
        const double dt_ = zSz / d;
        const double prev = wv::load_uniform(wv::lds(sc + 4));
        wv::store_uniform(wv::lds(sc + 0), dt_);
        wv::store_uniform(wv::lds(sc + 4), static_diffusion_update<d>(fixed_diffusion, success_iter, prev, dt_));
      }
    )
    ODEF_TILES_PHASE()  // the tile threads wait here for the helper
    ODEF_STAMP(9)
    // rows of L1 times Q (src/filtering.jl:85-89): one thread per row, the row in registers
    ODEF_TILES_PHASE(
      if (tid < D) {
        double w[d2];
_Pragma("unroll")
        for (int c = 0; c < d2; ++c) w[c] = L1[tid * W::LDL + c];
        static_for<0, d>([&](auto kc_) {
          constexpr int k = decltype(kc_)::value;
          double s = 0.0;
_Pragma("unroll")
          for (int c = k; c < d2; ++c) s += w[c] * HV[k * d2 + c];
          s *= beta[k];
_Pragma("unroll")
          for (int c = k; c < d2; ++c) w[c] -= s * HV[k * d2 + c];
        });
        double s = mp[tid];
_Pragma("unroll")
        for (int r = 0; r < d; ++r) s -= w[r] * y[r];
        m[tid] = tab[kTabPIJ + tid / d] * s;  // un-precondition (src/perform_step.jl:75)
_Pragma("unroll")
        for (int r = 0; r < d; ++r) ZP[tid * W::LDZ + r] = w[d + r];
      }
    )
    ODEF_STAMP(10)
    // Sigma_filt = Zp Zp' + Schur (own tile), un-preconditioned
    ODEF_TILES_PHASE(
      if (S.I >= 0) {
        const bool schur = S.J >= d2 / TS;
        double acc[TS][TS];
        _Pragma("unroll")
        for (int r = 0; r < TS; ++r)
          _Pragma("unroll")
          for (int c = 0; c < TS; ++c) acc[r][c] = schur ? S.x[r][c] : 0.0;
        const double* zi = ZP + (S.I * TS) * W::LDZ;
        const double* zj = ZP + (S.J * TS) * W::LDZ;
        for (int kk = 0; kk < d; ++kk) {
          double a_[TS];
          double b_[TS];
          _Pragma("unroll")
          for (int r = 0; r < TS; ++r) {
            a_[r] = zi[r * W::LDZ + kk];
            b_[r] = zj[r * W::LDZ + kk];
          }
          _Pragma("unroll")
          for (int r = 0; r < TS; ++r)
            _Pragma("unroll")
            for (int c = 0; c < TS; ++c) acc[r][c] += a_[r] * b_[c];
        }
        const double pipi = tab[kTabPIPI + (S.I / tpb) * MAXNB + (S.J / tpb)];
        _Pragma("unroll")
        for (int r = 0; r < TS; ++r)
          _Pragma("unroll")
          for (int c = 0; c < TS; ++c) S.x[r][c] = acc[r][c] * pipi;
      }
    )
    ODEF_STAMP(11)
  }

  // tile ownership, congruence coefficients, zero covariance, Taylor-mode initial mean (src/state_initialization.jl)
  template <bool HELPER>
  ODEF_TILES_FN void setup(const FilterParams& P, long i, const double* pl, int tid_dev, double* __restrict__ sm,
                                      TileState* st) {
    (void)tid_dev;
    double* m = sm + W::MV;
    double* sc = sm + W::SC;
    const size_t N = (size_t)P.N;
    ODEF_TILES_PHASE(
      S.I = -1;
      S.J = -1;
      S.tile = tid;
      if (tid < W::ntiles) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= tid) ++I;
        S.I = I;
        S.J = tid - I * (I + 1) / 2;
      }
      for (int j = 0; j < MAXNB; ++j) {
        S.aj[j] = 0.0;
        S.ak[j] = 0.0;
      }
      S.qjk = 0.0;
      if (S.I >= 0) {
        for (int Jb = 0; Jb < NB; ++Jb)
          for (int j = 0; j < NB; ++j) {
            if (Jb == S.I / tpb) S.aj[j] = P.pc.At[Jb][j];
            if (Jb == S.J / tpb) S.ak[j] = P.pc.At[Jb][j];
            if (Jb == S.I / tpb && j == S.J / tpb) S.qjk = P.pc.Qt[Jb][j];
          }
      }
      _Pragma("unroll")
      for (int r = 0; r < TS; ++r)
        _Pragma("unroll")
        for (int c = 0; c < TS; ++c) S.x[r][c] = 0.0;
      if (tid == 0) {
        double u0[d];
        double m0[D];
        for (int a = 0; a < d; ++a) u0[a] = P.u0[(size_t)a * N + i];
        taylor_init<RHS, q>(u0, pl, m0);
        for (int k = 0; k < D; ++k) m[k] = m0[k];
        for (int k = 0; k < 8; ++k) sc[k] = 0.0;
        for (int a = 0; a < d; ++a) sm[W::UC + a] = u0[a];
      }
    )
  }

  // one saved record: mean, packed covariance (own tile), diffusion
  template <bool HELPER>
  ODEF_TILES_FN void save_record(const FilterParams& P, long i, long slot, double diffusion, int tid_dev,
                                            double* __restrict__ sm, TileState* st) {
    (void)tid_dev;
    const double* m = sm + W::MV;
    const size_t N = (size_t)P.N;
    ODEF_TILES_PHASE(
      if (tid < D) P.mean[((size_t)slot * D + tid) * N + i] = m[tid];
      if (S.I >= 0) {
        _Pragma("unroll")
        for (int r = 0; r < TS; ++r)
          _Pragma("unroll")
          for (int c = 0; c < TS; ++c) {
            const int a = S.I * TS + r;
            const int b = S.J * TS + c;
            if (b <= a) P.cov[((size_t)slot * TRI + tri(a, b)) * N + i] = S.x[r][c];
          }
      }
      if (tid == 0) P.diff[(size_t)slot * N + i] = diffusion;
    )
  }

  // whole fixed-step solve of trajectory i.  `st`: one TileState (device) / kTilesThreads of them (host).
  template <bool HELPER>
  ODEF_TILES_FN void run(const FilterParams& P, long i, int tid_dev, double* __restrict__ sm, TileState* st) {
    double* m = sm + W::MV;
    double* sc = sm + W::SC;
    const size_t N = (size_t)P.N;
    __attribute__((unused)) double pl_local[RHS::np > 0 ? RHS::np : 1];
    const double* pl = pl_local;
    for (int k = 0; k < RHS::np; ++k) pl_local[k] = P.p_shared ? P.p[k] : P.p[(size_t)k * N + i];
    setup<HELPER>(P, i, pl, tid_dev, sm, st);
    if (P.everystep) save_record<HELPER>(P, i, 0, 0.0, tid_dev, sm, st);
    for (long n = 0; n < P.nsteps; ++n) {
      const double* tab = P.ptab + (size_t)P.tab_idx[n] * kTabStride;
      step<HELPER>(P.pc, pl, tab, P.fixed_diffusion, (int)n, sm, st, tid_dev);
      if (P.everystep) save_record<HELPER>(P, i, n + 1, sc[4], tid_dev, sm, st);
    }
    if (!P.everystep) save_record<HELPER>(P, i, 0, sc[4], tid_dev, sm, st);
    ODEF_TILES_PHASE(
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
    )
  }

  // Adaptive solve of trajectory i: the lane kernel's loop (filter_adaptive_lane, ek_lane.h) with the scalar control
  // -- step size, error estimate (src/perform_step.jl:78-84,148-158), PI controller, accept/commit -- done by tile
  // thread 0, whose decisions reach everybody (the helper wavefront included) through LDS control words behind a
  // barrier, so the loop stays workgroup-uniform.  One record per ATTEMPTED step, as in the lane kernel: a rejected
  // attempt re-reads the previous record and writes it again at the unchanged time.
  template <bool HELPER>
  ODEF_TILES_FN void run_adaptive(const FilterParams& P, long i, int tid_dev, double* __restrict__ sm, TileState* st) {
    double* m = sm + W::MV;
    double* sc = sm + W::SC;
    double* tabL = sm + W::TAB;
    double* wd = sm + W::WD;
    double* ucur = sm + W::UC;
    double* ctl = sm + W::CTL;  // [0] go on, [1] restore the previous state, [2] slot of this attempt's record, [3] naccept,
                                // [4] state finite
    const size_t N = (size_t)P.N;
    __attribute__((unused)) double pl_local[RHS::np > 0 ? RHS::np : 1];
    const double* pl = pl_local;
    for (int k = 0; k < RHS::np; ++k) pl_local[k] = P.p_shared ? P.p[k] : P.p[(size_t)k * N + i];
    setup<HELPER>(P, i, pl, tid_dev, sm, st);
    save_record<HELPER>(P, i, 0, 0.0, tid_dev, sm, st);
    // controller state: meaningful in tile thread 0 only
    const Controller& ct = P.ctrl;
    double t = P.t0, h = P.dt0, qold = ct.qoldinit, q11 = 1.0, sc3_prev = 0.0, sc4_prev = 0.0;
    int naccept = 0, nreject = 0, nsaved = 1, ret = 0;
    long attempts = 0;
    const long max_attempts = 20 * P.max_save + 1000;
    ODEF_TILES_PHASE(
      if (tid == 0) P.tsave[i] = P.t0;
    )
    for (;;) {
      ODEF_TILES_PHASE(
        if (tid == 0) {
          double go = 1.0;
          if (!(t < P.t1)) go = 0.0;
          else if (nsaved >= P.max_save || attempts >= max_attempts) { ret = 1; go = 0.0; }  // MaxIters
          else {
            ++attempts;
            h = fmin(h, ct.dtmax);
            h = fmin(h, P.t1 - t);  // tstop clipping
            if (!(h > ct.dtmin)) { ret = 2; go = 0.0; }  // DtLessThanMin
            else {
              precond_fill<NB>(h, precond_val<q>(h), tabL);
              sc3_prev = sc[3];
              sc4_prev = sc[4];
            }
          }
          ctl[0] = go;
          ctl[3] = (double)naccept;
        }
      )
      if (ctl[0] == 0.0) break;  // workgroup-uniform
      step<HELPER>(P.pc, pl, tabL, P.fixed_diffusion, (int)ctl[3], sm, st, tid_dev);
      ODEF_TILES_PHASE(
        if (tid == 0) {
          // DiffEqBase.calculate_residuals! + ODE_DEFAULT_NORM (src/perform_step.jl:78-84); sc[0] = local diffusion
          double acc = 0.0;
          for (int r = 0; r < d; ++r) {
            const double es = sqrt(sc[0] * wd[r]);
            const double e = h * es / (P.abstol + fmax(fabs(ucur[r]), fabs(m[r])) * P.reltol);
            acc += e * e;
          }
          double EEst = sqrt(acc / d);
          if (!(EEst == EEst) || !(fabs(EEst) <= 1.79769313486231570815e+308)) EEst = INFINITY;
          for (int r = 0; r < d; ++r) ucur[r] = m[r];  // integ.u .= u_filt, also when rejected (src/perform_step.jl:86)
          double qq;
          if (EEst == 0.0) {
            qq = 1.0 / ct.qmax;
          } else {
            q11 = pow(EEst, ct.beta1);
            qq = q11 / pow(qold, ct.beta2);
            qq = fmax(1.0 / ct.qmax, fmin(1.0 / ct.qmin, qq / ct.gamma));
          }
          const bool accepted = EEst <= 1.0;  // OrdinaryDiffEq accepts on <=, the cache commits on < (:89)
          const bool restore = !(EEst < 1.0);
          if (restore) {
            sc[3] = sc3_prev;
            sc[4] = sc4_prev;
          }
          if (accepted) {
            if (qq <= ct.qsteady_max && qq >= ct.qsteady_min) qq = 1.0;
            qold = fmax(EEst, ct.qoldinit);
            double tn = t + h;
            if (fabs(tn - P.t1) < 100.0 * 2.220446049250313e-16 * fmax(fabs(tn), fabs(P.t1))) tn = P.t1;
            t = tn;
            ++naccept;
            h = h / qq;
          } else {
            ++nreject;
            h = h / fmin(1.0 / ct.qmin, q11 / ct.gamma);
          }
          ctl[1] = restore ? 1.0 : 0.0;
          ctl[2] = (double)nsaved;
          P.tsave[(size_t)nsaved * N + i] = t;
        }
      )
      const long slot = (long)ctl[2];
      if (ctl[1] != 0.0) {  // x_filt is not committed: back to the previous record, cache.x = P^-1 (P x) (:73)
        ODEF_TILES_PHASE(
          if (tid < D) {
            const double v = P.mean[((size_t)(slot - 1) * D + tid) * N + i];
            m[tid] = tabL[kTabPIJ + tid / d] * (tabL[kTabPJ + tid / d] * v);
          }
          if (S.I >= 0) {
            _Pragma("unroll")
            for (int r = 0; r < TS; ++r)
              _Pragma("unroll")
              for (int c = 0; c < TS; ++c)
                S.x[r][c] = P.cov[((size_t)(slot - 1) * TRI + symidx(S.I * TS + r, S.J * TS + c)) * N + i];
          }
        )
      }
      save_record<HELPER>(P, i, slot, sc[4], tid_dev, sm, st);
      bool finite = true;
      ODEF_TILES_PHASE(
        if (tid == 0) {
          ++nsaved;
          for (int k = 0; k < D; ++k) finite = finite && (fabs(m[k]) <= 1.79769313486231570815e+308);
          if (!finite) ret = 3;
          ctl[4] = finite ? 1.0 : 0.0;  // its own word: ctl[0] is rewritten by thread 0 right after this barrier
        }
      )
      if (ctl[4] == 0.0) break;
    }
    ODEF_TILES_PHASE(
      if (tid == 0) {
        P.loglik[i] = sc[3];
        P.naccept[i] = naccept;
        P.nreject[i] = nreject;
        P.nf[i] = naccept + nreject;
        P.njac[i] = IS_EK1 ? naccept + nreject : 0;
        P.nsaved[i] = nsaved;
        P.retcode[i] = ret;
      }
    )
  }
};

}  // namespace odef
