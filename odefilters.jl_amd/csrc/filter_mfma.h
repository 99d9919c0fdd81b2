// EK0/EK1 fixed-step filter for large state dimension (Pleiades: d = 28, D = 168) on the FP64 matrix cores: one
// 576-thread workgroup per trajectory, the covariance DISTRIBUTED IN REGISTERS as 16 x 16 accumulator tiles of
// v_mfma_f64_16x16x4_f64 for the whole solve, LDS (<= 160 KB) as exchange medium and operand store.
//
// Formulation (src/perform_step.jl:27-93, src/filtering.jl:33-48,79-91): the same Gaussian conditioning the reference
// performs on square roots, written on the covariance itself in the JOSEPH form
//     S^- = A S A' + sigma2 Q                          (Kronecker structure: A = At (x) I_d)
//     C   = S^- H',   Sm = H C = L L',   W = L^-1       (d x d, the only factorisation of the step)
//     V   = C W',     K = V W,           m = m^- - V (W z)
//     T   = S^- - V V'                                  ( = (I - K H) S^- , symmetric by construction)
//     E   = T H'                                        ( = 0 in exact arithmetic: the rounding residual of T)
//     S   = T - E K'                                    ( = (I - K H) S^- (I - K H)' )
// With R = 0 the plain form S^- - K Sm K' amplifies rounding errors step after step (tests/golden/make_exact.py); the
// second stage removes exactly that residual.  In float64 this recursion stays as close to the extended-precision
// evaluation of the reference algorithm as the reference's own square-root arithmetic.
// Every D-sized operation is a rank-d product -- 2 d D^2 flops each, on the matrix pipe -- and there is no D-sized
// factorisation, no Householder QR and no per-pivot synchronisation of the workgroup: the step has 11 barriers instead
// of ~45 (filter_tiles.h, which stays in the library: adaptive solves, ODEF_PLEIADES_FILTER=tiles).
//
// Tiles.  Each derivative block (d = 28 rows) is split into two tiles of 14 real rows + 2 zero rows, so that the
// Kronecker congruence maps whole tiles onto whole tiles (tile 2 b + h <-> rows 14 h .. 14 h + 13 of block b) and
// DP = 32 (q + 1) = 192 at order 5.  Only tiles (Q, P) with Q <= P are kept (78 of them), 9-10 per wavefront, in the
// accumulator layout
//     register v of lane l  <->  element (row 4 v + l / 16, column l % 16).
// A tile in this layout IS the B operand of a K = 16 product (register v = k-step v), so  H (.) tile  needs no data
// movement; products that need the tile as the A operand (the six tiles above the diagonal of the first two derivative
// blocks) go through a small LDS copy.  The rank-d updates read their D x d operand panels (V, K, E) from LDS.
//
// Code.  Eight tile wavefronts share ONE copy of the tile code (unrolled over the <= 10 tile slots of a wavefront, the
// tile coordinates of a slot in SGPRs); the ninth wavefront (the helper) owns no tile and has its own code path: the two
// d x d factorisations -- H Q H' for sigma^2 beside the congruence of the others (sigma^2 Q is added to the tiles and
// sigma^2 Q H' to the panel only after C0 = (A S A') H' is there), Sm while everybody waits -- and, beside the tail of
// step n, the whole measurement chain of step n + 1 (mean prediction, f, J, z, H0, M0, H Q H', the padded H0'), which
// depends on the mean alone.  The step body has to stay near the 64 KB of the instruction cache: see Slots below.
#pragma once
#ifndef ODEF_HOST_EMUL
#include "ek_lane.h"
#include "mfma_dense.h"
#include "team.h"
#include "wave_vec.h"

namespace odef {

// The same map for the workgroup-per-trajectory kernels (D = 168): consecutive trajectories sit next to each other in every
// record row ([field][N] layout: 8 bytes apart), so the 8-byte pieces a workgroup stores or loads share their 64-byte sector with
// the 7 neighbouring trajectories.  With the plain map those neighbours run on 8 different XCDs -- 8 L2s each holding one piece
// of the sector, every piece a partial write / a whole sector fetched for 8 bytes.  Giving each XCD a contiguous range of
// trajectories puts the neighbours into ONE L2 at about the same time, where the pieces merge.  Grid: team_grid(N) workgroups;
// returns -1 for the padding workgroups.
inline unsigned team_grid(long N) { return (unsigned)((N + 7) / 8 * 8); }
__device__ inline long team_traj(long N) {
  const long per = (long)(gridDim.x / 8u);
  const long i = (long)(blockIdx.x % 8u) * per + (long)(blockIdx.x / 8u);
  return i < N ? i : -1;
}

constexpr int kMfWaves = 9;       // wavefronts of the workgroup
constexpr int kMfTileWaves = 8;   // ... that own tiles; the last one is the helper
constexpr int kMfHelper = 8;
constexpr int kMfBlock = 64 * kMfWaves;
#ifndef ODEF_MF_HELPER_BIAS
#define ODEF_MF_HELPER_BIAS 0
#endif
constexpr int kMfHelperBias = ODEF_MF_HELPER_BIAS;  // tiles fewer for the tile wavefronts that share the helper's SIMD (measured: no gain)

// In-kernel stamps at phase boundaries: ONLY in the diagnostic build of tools/mfma_filter_stamps.hip
// (-DODEF_MF_STAMPS); the product kernel contains none.  STAMP: tile thread 0; HSTAMP: the helper's own clock.
#ifdef ODEF_MF_STAMPS
__device__ unsigned long long* g_mf_stamp_buf = nullptr;
__device__ unsigned long long g_mf_stamp_last = 0;
__device__ unsigned long long g_mf_hstamp_last = 0;
#define ODEF_MF_STAMP(k)                                                      \
  if (tid == 0 && blockIdx.x == 0 && g_mf_stamp_buf) {                        \
    unsigned long long t_;                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    g_mf_stamp_buf[(k)] += t_ - g_mf_stamp_last;                              \
    g_mf_stamp_last = t_;                                                     \
  }
#define ODEF_MF_HSTAMP(k)                                                     \
  if (threadIdx.x == 64 * kMfHelper && blockIdx.x == 0 && g_mf_stamp_buf) {   \
    unsigned long long t_;                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    g_mf_stamp_buf[16 + (k)] += t_ - g_mf_hstamp_last;                        \
    g_mf_hstamp_last = t_;                                                    \
  }
#else
#define ODEF_MF_STAMP(k)
#define ODEF_MF_HSTAMP(k)
#endif

// Tile ownership.  The "head" of a tile column (its tiles in the first four tile rows: the ones H (.) reads) stays in
// one wavefront, so that C = S^- H' needs no reduction across wavefronts; every other tile may go anywhere.  Heads are
// dealt first (largest column first, to the wavefront with most room), then the remaining tiles column by column,
// staying with a wavefront until it is full (the B fragments of a column are then reloaded rarely).
// 78 tiles over 8 wavefronts: 9, 9, 10, 10, 10, 10, 10, 10.
template <int NT>
struct MfOwnTab {
  static constexpr int ntiles = NT * (NT + 1) / 2;
  static constexpr int cap = (ntiles + kMfTileWaves - 1) / kMfTileWaves + 4;  // upper bound of a wavefront's load
  int n[kMfTileWaves];
  int Q[kMfTileWaves][cap];
  int P[kMfTileWaves][cap];
  bool ok;
};
template <int NT>
constexpr MfOwnTab<NT> make_mf_own() {
  MfOwnTab<NT> t{};
  constexpr int cap = MfOwnTab<NT>::cap, ntiles = MfOwnTab<NT>::ntiles;
  t.ok = true;
  for (int w = 0; w < kMfTileWaves; ++w) {
    t.n[w] = 0;
    for (int s = 0; s < cap; ++s) {
      t.Q[w][s] = -1;
      t.P[w][s] = -1;
    }
  }
  int tgt[kMfTileWaves] = {};
  {
    int nb = 0;
    for (int w = 0; w < kMfTileWaves; ++w) nb += ((w & 3) == (kMfHelper & 3));
    const int total = ntiles + nb * kMfHelperBias;
    for (int w = 0; w < kMfTileWaves; ++w) tgt[w] = total / kMfTileWaves - (((w & 3) == (kMfHelper & 3)) ? kMfHelperBias : 0);
    int rest = ntiles;
    for (int w = 0; w < kMfTileWaves; ++w) rest -= tgt[w];
    for (int w = kMfTileWaves - 1; rest > 0; --w, --rest) tgt[w < 0 ? 0 : w] += 1;
  }
  auto roomiest = [&]() {
    int w = 0;
    for (int k = 1; k < kMfTileWaves; ++k)
      if (tgt[k] - t.n[k] > tgt[w] - t.n[w]) w = k;
    return w;
  };
  auto put = [&](int w, int Q, int P) {
    if (t.n[w] >= cap) {
      t.ok = false;
      return;
    }
    t.Q[w][t.n[w]] = Q;
    t.P[w][t.n[w]] = P;
    ++t.n[w];
  };
  for (int P = NT - 1; P >= 0; --P) {  // heads
    const int w = roomiest();
    for (int Q = 0; Q <= (P < 3 ? P : 3); ++Q) put(w, Q, P);
  }
  int cur = roomiest();
  for (int P = NT - 1; P >= 4; --P)
    for (int Q = 4; Q <= P; ++Q) {
      if (t.n[cur] >= tgt[cur]) cur = roomiest();
      put(cur, Q, P);
    }
  return t;
}

template <int d, int NB>
struct MfLds {
  static constexpr int D = d * NB, NT = 2 * NB, DP = 16 * NT, ntiles = NT * (NT + 1) / 2;
  static constexpr int TR = d / 2, TSZ = TR * TR;  // real rows per tile; one tile in the exchange
  static constexpr int LDP = 34;                   // pitch of the operand panels (32 + 2: conflict-light fragment reads)
  // region R0, time-shared: tile exchange of the congruence  |  panels V / E, K + the transposed-role tile copies
  static constexpr int EX = 0, EX_size = ntiles * TSZ;
  static constexpr int VP = 0, KP = VP + DP * LDP, TL = KP + DP * LDP;  // TL: 6 tiles above the diagonal + 4 diagonal ones
  static constexpr int R0_need = TL + 10 * 256;
  static constexpr int R0_size = EX_size > R0_need ? EX_size : R0_need;
  static constexpr int DUMMY = EX_size;  // 64 doubles behind the exchange: where the padding lanes of a tile store go
  static_assert(R0_size >= EX_size + 64, "room for the dummy stores");
  // region B, time-shared:
  //   H0 | M0 | WM          raw / scaled Jacobian block, M0, W = H Q H'           (measurement chain ... chol(W), Sm)
  //   scratch of chol(W)    over H0 | M0 once W is built: two diagonal blocks (+ pivots), the block below, L21, W11, W22
  //   WL | scratch of Sm    W_S = L^-1 of the innovation covariance [32][LDP], then the blocks of Sm and L21
  static constexpr int LDd = d + 1;
  static constexpr int CW11 = R0_size, CW22 = CW11 + 272, CW21 = CW22 + 272, CWL = CW21 + 256, CWW1 = CWL + 256, CWW2 = CWW1 + 256;
  // W sits behind H0 | M0 AND behind the scratch of its own factorisation (six 16 x 16 blocks, sized for d <= 32 whatever d is:
  // at d = 28 the two coincide, a smaller d leaves a gap)
  static constexpr int H0 = R0_size, M0 = H0 + d * d, WM = (M0 + d * d > CWW2 + 256) ? M0 + d * d : CWW2 + 256;
  static_assert(CWW2 + 256 <= WM, "the scratch of chol(W) must not reach W itself");
  static constexpr int WL = R0_size, SB11 = WL + 32 * LDP, SB22 = SB11 + 272, SB21 = SB22 + 272, L21 = SB21 + 256;
  static constexpr int B_need1 = (WM - R0_size) + d * LDd, B_need2 = L21 + 256 - R0_size;
  static constexpr int B_size = B_need1 > B_need2 ? B_need1 : B_need2;
  static constexpr int HS0 = R0_size + B_size;  // [32][LDP]: H0' in padded indices (k = state column, a = measurement)
  static constexpr int MV = HS0 + 32 * LDP, MT = MV + D, MP = MT + D, Z = MP + D, YV = Z + 32, UP = YV + 32;  // z: d values + zeros up to 32
  static constexpr int DU = UP + d, WD = DU + d, UC = WD + d, SC = UC + d, CTL = SC + 8, TAB = CTL + 8;
  static constexpr int CF1 = TAB + kTabStride, CF2 = CF1 + NB * NB * NB;  // coefficients of the two congruence stages
  static constexpr int size = CF2 + NB * NB;
  static_assert(d % 2 == 0 && d / 2 <= 16 && d / 2 >= 1, "a derivative block is split into two tiles of d / 2 <= 16 rows");
  static_assert(size * 8 <= 160 * 1024, "LDS budget of one workgroup");
};

template <class RHS, int q, bool IS_EK1>
struct MfmaFilter {
  static constexpr int d = RHS::d, NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  using W = MfLds<d, NB>;
  using d4 = mf::d4;
  static constexpr int NT = W::NT, TR = W::TR, TSZ = W::TSZ, LDP = W::LDP, NTHR = kMfBlock;
  static constexpr int KS = (d + 3) / 4;  // k-steps of a product over the measurement index
  static_assert(make_mf_own<NT>().ok, "tile ownership table overflow");
  static constexpr int max_load() {
    int mx = 0;
    for (int w = 0; w < kMfTileWaves; ++w)
      if (make_mf_own<NT>().n[w] > mx) mx = make_mf_own<NT>().n[w];
    return mx;
  }
  static constexpr int NS = max_load();  // tile slots per wavefront
  // The tile code exists ONCE (shared by the eight tile wavefronts, unrolled over the NS slots); which tile sits in slot s
  // of a wavefront is a wave-uniform run-time fact (SGPRs).  A per-wavefront specialisation (compile-time tile lists)
  // executes 5x fewer instructions per wavefront but is eight instruction streams of 17 KB each: with the helper's code
  // that is 190 KB against a 64 KB instruction cache, and measured 1.6x SLOWER.  What matters is that every per-slot
  // address is "scalar base + lane part": the lane parts are computed once per phase, the bases on the scalar unit.
  struct Slots {
    int n;               // tiles of this wavefront
    int tq[NS], tp[NS];  // tile coordinates of slot s (valid for s < n)
  };

  struct Geo {  // per-lane geometry of the accumulator layout
    int g, j, lane;
    bool ok[4];  // element (4 v + g, j) is a real entry of its tile
    int sym[4];  // offset of element (min, max) of (4 v + g, j) in a compact TR x TR tile: upper-triangle read of a diagonal tile
  };
  // The lane geometry is loop-invariant, and so is every LDS address derived from it: left alone, the compiler hoists a few
  // hundred of them out of the step loop, runs out of registers and reloads them from scratch inside the phases.  Each
  // phase therefore starts from a laundered copy -- no instructions, the values just stop being provably invariant.
  __device__ __attribute__((always_inline)) static inline Geo fresh(const Geo& G0) {
    Geo G;
    int g = G0.g, j = G0.j, lane = G0.lane;
    asm volatile("" : "+v"(g), "+v"(j), "+v"(lane));
    G.g = g;
    G.j = j;
    G.lane = lane;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = 4 * v + g;
      G.ok[v] = (i < TR) && (j < TR);
      G.sym[v] = (i < j ? i : j) * TR + (i < j ? j : i);
    }
    return G;
  }

#define ODEF_MF_FN __device__ __attribute__((always_inline)) static inline

  ODEF_MF_FN constexpr int pad_d(int a) { return 16 * (a / TR) + a % TR; }  // measurement / in-block index -> padded
  ODEF_MF_FN constexpr int uidx(int Q, int P) { return P * (P + 1) / 2 + Q; }

  // ------------------------------------------------------------------------------------------------ tile phases
  // tiles -> compact exchange slots; the padding lanes of a tile store into a dummy area (no exec masking)
  ODEF_MF_FN void ex_put_all(const d4 (&T)[NS], const Slots& S, const Geo& G0, double* __restrict__ ex) {
    const Geo G = fresh(G0);
    int rel[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) rel[v] = G.ok[v] ? (4 * v + G.g) * TR + G.j : -1;
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (s < S.n) {
        const int base = uidx(S.tq[s], S.tp[s]) * TSZ;
#pragma unroll
        for (int v = 0; v < 4; ++v) ex[rel[v] >= 0 ? base + rel[v] : W::DUMMY + G.lane] = T[s][v];
      }
    });
  }

  // first stage of predict_cov! (src/filtering.jl:33-41):  Z(Q, P) = sum_{k >= Q/2} At[Q/2][k] pj[k] pj[P/2] S(2k + Q%2, P),
  // sources in the own tile COLUMN; those below the diagonal are read transposed (S symmetric), a diagonal one through
  // its upper triangle (the rank updates leave rounding-level asymmetry there).  One LDS round trip per source tile:
  // its four values and the next coefficient.
  ODEF_MF_FN void stage1(d4 (&T)[NS], const Slots& S, const Geo& G0, const double* __restrict__ ex, const double* __restrict__ cf1) {
    const Geo G = fresh(G0);
    const double* dir = ex + G.g * TR + G.j;
    const double* tra = ex + G.j * TR + G.g;
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (s < S.n) {
        const int Q = S.tq[s], P = S.tp[s], a = Q >> 1, hq = Q & 1, b = P >> 1;
        const double* cf = cf1 + (a * NB + b) * NB;
        d4 acc = mf::zero4();
        double ck = cf[a];
#pragma nounroll
        for (int k = a; k < NB; ++k) {
          const int Qs = 2 * k + hq;
          const double cn = cf[k + 1 < NB ? k + 1 : k];
          double x[4];
          if (Qs < P) {
            const double* src = dir + uidx(Qs, P) * TSZ;
#pragma unroll
            for (int v = 0; v < 4; ++v) x[v] = src[4 * v * TR];
          } else if (Qs == P) {
            const double* src = ex + uidx(Qs, P) * TSZ;
#pragma unroll
            for (int v = 0; v < 4; ++v) x[v] = src[G.sym[v]];
          } else {
            const double* src = tra + uidx(P, Qs) * TSZ;
#pragma unroll
            for (int v = 0; v < 4; ++v) x[v] = src[4 * v];
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] += ck * x[v];
          ck = cn;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) T[s][v] = G.ok[v] ? acc[v] : 0.0;
      }
    });
  }
  // second stage:  S^-(Q, P) = sum_{k >= P/2} At[P/2][k] Z(Q, 2k + P%2)   (sigma2 Qt is added later), sources in the own tile ROW
  ODEF_MF_FN void stage2(d4 (&T)[NS], const Slots& S, const Geo& G0, const double* __restrict__ ex, const double* __restrict__ cf2) {
    const Geo G = fresh(G0);
    const double* dir = ex + G.g * TR + G.j;
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (s < S.n) {
        const int Q = S.tq[s], P = S.tp[s], b = P >> 1, hp = P & 1;
        const double* cf = cf2 + b * NB;
        d4 acc = mf::zero4();
        double ck = cf[b];
#pragma nounroll
        for (int k = b; k < NB; ++k) {
          const double cn = cf[k + 1 < NB ? k + 1 : k];
          const double* src = dir + uidx(Q, 2 * k + hp) * TSZ;
          double x[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) x[v] = src[4 * v * TR];
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] += ck * x[v];
          ck = cn;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) T[s][v] = G.ok[v] ? acc[v] : 0.0;
      }
    });
  }
  // + sigma2 Q (src/filtering.jl:35): Qt[Q/2][P/2] on the diagonal of the tiles with equal halves
  ODEF_MF_FN void add_sigma2_q(d4 (&T)[NS], const Slots& S, const Geo& G0, const PriorConsts& pc, double sigma2) {
    const Geo G = fresh(G0);
    bool dg[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) dg[v] = G.ok[v] && 4 * v + G.g == G.j;
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (s < S.n) {
        const int Q = S.tq[s], P = S.tp[s];
        if ((Q & 1) == (P & 1)) {
          const double sq = sigma2 * pc.Qt[Q >> 1][P >> 1];
#pragma unroll
          for (int v = 0; v < 4; ++v) T[s][v] += dg[v] ? sq : 0.0;
        }
      }
    });
  }

  // the ten tiles of the first four tile columns, as full 16 x 16 row-major copies: the six above the diagonal (read
  // back transposed by hproject) and the four diagonal ones (read back through their upper triangle by diag_resym)
  ODEF_MF_FN void tl_put(const d4 (&T)[NS], const Slots& S, const Geo& G0, double* __restrict__ tl, bool with_diag) {
    const Geo G = fresh(G0);
    double* dst0 = tl + G.g * 16 + G.j;
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (s < S.n) {
        const int Q = S.tq[s], P = S.tp[s];
        if (P <= 3 && (Q < P || with_diag)) {
          double* dst = dst0 + (Q < P ? P * (P - 1) / 2 + Q : 6 + Q) * 256;
#pragma unroll
          for (int v = 0; v < 4; ++v) dst[64 * v] = T[s][v];
        }
      }
    });
  }
  // A diagonal tile must be EXACTLY symmetric where it enters a product as a whole (H (.) tile): its antisymmetric part
  // is not damped by the update but multiplied by (I + K H), step after step (numpy model in DESIGN.md section 3.9).
  ODEF_MF_FN void diag_resym(d4 (&T)[NS], const Slots& S, const Geo& G0, const double* __restrict__ tl) {
    const Geo G = fresh(G0);
    int so[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = 4 * v + G.g;
      so[v] = (i < G.j ? i : G.j) * 16 + (i < G.j ? G.j : i);
    }
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (s < S.n) {
        const int Q = S.tq[s], P = S.tp[s];
        if (Q == P && P <= 3) {
          const double* src = tl + (6 + Q) * 256;
#pragma unroll
          for (int v = 0; v < 4; ++v) T[s][v] = src[so[v]];
        }
      }
    });
  }

  // (C_P)' = Hs' (.) over the tiles of the first two derivative blocks of every tile column, panel OUT[row][a]:
  //   block 0 (tiles Q = 0, 1):  H0-part, 8 MFMAs per tile;  block 1 (tiles 2, 3):  h1 I, one FMA per element.
  // The panel is indexed by the PLAIN measurement index a (28 real columns + 4 zero ones: 7 k-steps in the rank updates
  // instead of 8): row i of accumulator ta is a = 14 ta + i for i < 14; its two zero rows go to 28 + 2 ta + (i - 14).
  ODEF_MF_FN void hproject(const d4 (&T)[NS], const Slots& S, const Geo& G0, const double* __restrict__ hs0,
                           const double* __restrict__ tl, double h1, double* __restrict__ out) {
    const Geo G = fresh(G0);
    const double* hp = hs0 + G.g * LDP + G.j;
    const double* tt = tl + G.j * 16 + G.g;
    int oa[4], ob[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = 4 * v + G.g;
      oa[v] = G.j * LDP + (i < TR ? i : 2 * TR + (i - TR));
      ob[v] = G.j * LDP + (i < TR ? TR + i : 2 * TR + 2 + (i - TR));
    }
    d4 acc0 = mf::zero4(), acc1 = mf::zero4();
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (s < S.n) {
        const int Q = S.tq[s], P = S.tp[s];
        if (Q <= 3) {  // only the head of a column enters
          if (Q <= 1) {
            const double* h = hp + 16 * Q * LDP;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              acc0 = mf::mfma(h[4 * ks * LDP], T[s][ks], acc0);
              acc1 = mf::mfma(h[4 * ks * LDP + 16], T[s][ks], acc1);
            }
          } else if (Q == 2) {
            acc0 += h1 * T[s];
          } else {
            acc1 += h1 * T[s];
          }
          if (Q == (P < 3 ? P : 3)) {  // last head tile of the column: the sources ABOVE the diagonal come transposed from the LDS copies
#pragma nounroll
            for (int Qc = P + 1; Qc <= 3; ++Qc) {
              const double* tq = tt + (Qc * (Qc - 1) / 2 + P) * 256;
              if (Qc == 1) {
                const double* h = hp + 16 * LDP;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                  const double b = tq[4 * ks];
                  acc0 = mf::mfma(h[4 * ks * LDP], b, acc0);
                  acc1 = mf::mfma(h[4 * ks * LDP + 16], b, acc1);
                }
              } else {
                d4 x;
#pragma unroll
                for (int v = 0; v < 4; ++v) x[v] = h1 * tq[4 * v];
                if (Qc == 2) acc0 += x; else acc1 += x;
              }
            }
            double* o = out + 16 * P * LDP;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              o[oa[v]] = acc0[v];
              o[ob[v]] = acc1[v];
            }
            acc0 = mf::zero4();
            acc1 = mf::zero4();
          }
        }
      }
    });
  }

  // T[s] -= A_panel[tile row Q] B_panel[tile row P]'  (rank-d update of every tile), then T[s] *= scale[Q/2][P/2] if given
  ODEF_MF_FN void rank_update(d4 (&T)[NS], const Slots& S, const Geo& G0, const double* __restrict__ ap,
                              const double* __restrict__ bp, const double* __restrict__ scale) {
    const Geo G = fresh(G0);
    const double* a0 = ap + G.j * LDP + G.g;
    const double* b0 = bp + G.j * LDP + G.g;
    double bf[KS];
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (s < S.n) {
        const int Q = S.tq[s], P = S.tp[s];
        bool reload = true;
        if constexpr (s > 0) reload = S.tp[s - 1] != P;
        if (reload) {  // first tile of a run of one column: its B fragments (negated: the product is subtracted)
          const double* b = b0 + 16 * P * LDP;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) bf[ks] = -b[4 * ks];
        }
        const double* a = a0 + 16 * Q * LDP;
        d4 acc = T[s];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mf::mfma(a[4 * ks], bf[ks], acc);
        if (scale) acc *= scale[(Q >> 1) * MAXNB + (P >> 1)];
        T[s] = acc;
      }
    });
  }

  // ------------------------------------------------------------------------------------------------ helper pieces
  // Cholesky of a 32 x 32 (28 real, plain index: 16 + 12) SPD matrix in two 16 x 16 blocks, by one wavefront:
  //   sb11 / sb22: the diagonal blocks (+ 16 reciprocal pivots behind each), sb21: the block below them (row-major [16][16])
  //   out: w11 / w22 = inverses of the two diagonal factors (pitch ldw), l21 = L21 [16][16]
  // (in two halves, so that the helper can spread the factorisation of H Q H' over two barrier intervals)
  ODEF_MF_FN void chol2_a(double* __restrict__ sb11, double* __restrict__ sb22, const double* __restrict__ sb21,
                          double* __restrict__ l21, double* __restrict__ w11, int ldw, const Geo& G) {
    mf::diag_block_factor(sb11, nullptr, w11, ldw);
    // L21 = S21 W11'
    d4 acc = mf::zero4();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = mf::mfma(sb21[G.j * 16 + 4 * ks + G.g], w11[G.j * ldw + 4 * ks + G.g], acc);
#pragma unroll
    for (int v = 0; v < 4; ++v) l21[(4 * v + G.g) * 16 + G.j] = acc[v];
    tv::lds_sync();
    // S22 <- S22 - L21 L21'
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = sb22[(4 * v + G.g) * 16 + G.j];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double f = l21[G.j * 16 + 4 * ks + G.g];
      acc = mf::mfma(-f, f, acc);
    }
    tv::lds_sync();
#pragma unroll
    for (int v = 0; v < 4; ++v) sb22[(4 * v + G.g) * 16 + G.j] = acc[v];
    tv::lds_sync();
  }
  ODEF_MF_FN void chol2_b(double* __restrict__ sb22, double* __restrict__ w22, int ldw) { mf::diag_block_factor(sb22, nullptr, w22, ldw); }
  // W = L^-1 of the innovation covariance into wl[32][LDP] (lower triangular)
  ODEF_MF_FN void factor_s(double* __restrict__ sb11, double* __restrict__ sb22, double* __restrict__ sb21,
                           double* __restrict__ l21, double* __restrict__ wl, const Geo& G) {
    for (int e = G.lane; e < 16 * 16; e += 64) wl[(e >> 4) * LDP + 16 + (e & 15)] = 0.0;  // the block above the diagonal
    chol2_a(sb11, sb22, sb21, l21, wl, LDP, G);
    chol2_b(sb22, wl + 16 * LDP + 16, LDP);
    // W21 = -W22 (L21 W11)
    d4 acc = mf::zero4();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = mf::mfma(l21[G.j * 16 + 4 * ks + G.g], wl[(4 * ks + G.g) * LDP + G.j], acc);
    d4 w21 = mf::zero4();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w21 = mf::mfma(-wl[(16 + G.j) * LDP + 16 + 4 * ks + G.g], acc[ks], w21);
#pragma unroll
    for (int v = 0; v < 4; ++v) wl[(16 + 4 * v + G.g) * LDP + G.j] = w21[v];
    tv::lds_sync();
  }
  ODEF_MF_FN double wave_sum(double x) {
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) x += __shfl_xor(x, msk, 64);
    return x;
  }

  // ---- the measurement chain of a step, by ONE wavefront (the helper), with wave-level LDS fences only: everything
  // that depends on the mean alone -- m^- = A P m, u_pred, f, J, z, H0, M0, W = H Q H', the padded H0' and the
  // coefficient tables of the congruence (src/perform_step.jl:36-43,95-132, src/diffusions.jl:78).  It runs for step
  // n + 1 beside the tail of step n, so that a step of the tile wavefronts starts with the congruence right away.
  ODEF_MF_FN void chain_a1(const PriorConsts& pc, const double* __restrict__ p, const double* __restrict__ tab, bool new_table,
                           double* __restrict__ sm, int lane) {
    double* H0 = sm + W::H0;
    double* WM = sm + W::WM;
    double* m = sm + W::MV;
    double* mt = sm + W::MT;
    double* mp = sm + W::MP;
    double* up = sm + W::UP;
    double* du = sm + W::DU;
    double* tabL = sm + W::TAB;  // the step's preconditioner table in LDS: per-lane indices into it below
    if (new_table) {  // on a uniform grid the table of the previous step is already there
      for (int e = lane; e < kTabStride; e += 64) tabL[e] = tab[e];
      tv::lds_sync();
    }
    for (int r = lane; r < D; r += 64) mt[r] = tabL[kTabPJ + r / d] * m[r];
    tv::lds_sync();
    const double pi0 = tabL[kTabPIJ + 0];
    for (int r = lane; r < D; r += 64) {
      const int J = r / d, a = r % d;
      double s = mt[r];
      for (int j = J + 1; j < NB; ++j) s += pc.At[J][j] * mt[j * d + a];
      mp[r] = s;
      if (r < d) up[r] = pi0 * s;
    }
    tv::lds_sync();
    if constexpr (HasWaveAssemble<RHS>::value) {
      // the vector field's own one-wavefront assembly: du, H0 = -J pi0 and M0 = H0 QL00 + I h1 QL10 in one go
      static_assert(RHS::team_scratch <= d * W::LDd, "pair buffer must fit the W area");
      RHS::team_eval_pairs(lane, up, WM);
      tv::lds_sync();
      const double h1 = tabL[kTabPIJ + 1];
      RHS::wave_assemble_scaled(lane, up, WM, du, H0, sm + W::M0, pi0, pc.QLt[0][0], h1 * pc.QLt[1][0], IS_EK1);
    } else if constexpr (HasTeamEval<RHS>::value) {
      static_assert(RHS::team_scratch <= d * W::LDd, "pair buffer must fit the W area");
      RHS::team_eval_pairs(lane, up, WM);
      tv::lds_sync();
      RHS::team_eval_assemble(lane, 64, up, WM, du, IS_EK1 ? H0 : nullptr);  // raw J into H0, scaled in chain_a2
    } else {
      if (lane == 0) {
        double u_[d], du_[d];
        for (int a = 0; a < d; ++a) u_[a] = up[a];
        RHS::f(u_, p, du_);
        for (int a = 0; a < d; ++a) du[a] = du_[a];
        if constexpr (IS_EK1) rhs_jacobian<RHS>(u_, p, *reinterpret_cast<double (*)[d][d]>(H0));  // (its own jac, else forward mode)
      }
    }
    tv::lds_sync();
  }
  ODEF_MF_FN void chain_a2(const PriorConsts& pc, double* __restrict__ sm, int lane) {
    double* H0 = sm + W::H0;
    double* M0 = sm + W::M0;
    const double* mp = sm + W::MP;
    const double* du = sm + W::DU;
    double* z = sm + W::Z;
    const double* tabL = sm + W::TAB;
    const double pi0 = tabL[kTabPIJ + 0], pi1 = tabL[kTabPIJ + 1], h1 = pi1;
    if (lane < d) z[lane] = pi1 * mp[d + lane] - du[lane];
    const double ql00 = pc.QLt[0][0], dg = h1 * pc.QLt[1][0];
    if constexpr (!HasWaveAssemble<RHS>::value)
    for (int r = lane >> 2; r < d; r += 16) {  // H0 = -J pi0 ; M0 = H0 QL00 + I h1 QL10 (src/diffusions.jl:78): row r, columns lane % 4 + 4 t
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        const int a = (lane & 3) + 4 * t;
        if (a < d) {
          double h0 = 0.0;
          if constexpr (IS_EK1) h0 = (0.0 - H0[r * d + a]) * pi0;
          H0[r * d + a] = h0;
          M0[r * d + a] = h0 * ql00 + (r == a ? dg : 0.0);
        }
      }
    }
    tv::lds_sync();
  }
  // W = M0 M0' + (h1 QL11)^2 I on the matrix pipe (lower blocks), diag(W) for the error estimate, coefficient tables
  ODEF_MF_FN void chain_b(const PriorConsts& pc, bool new_table, double* __restrict__ sm, const Geo& G) {
    const double* M0 = sm + W::M0;
    double* WM = sm + W::WM;
    const double* tabL = sm + W::TAB;
    const double h1 = tabL[kTabPIJ + 1], m1 = h1 * pc.QLt[1][1];
    // the fragments of the two row sets of M0 once (the three blocks use them as A and as B operand: the helper's stream is
    // private, every instruction of it an instruction-cache miss while the tile wavefronts run -- 14 loads instead of 42)
    double fr[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = 16 * t + G.j;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + G.g;
        fr[t][ks] = (r < d && k < d) ? M0[r * d + k] : 0.0;
      }
    }
    static_for<0, 3>([&](auto bc) {
      constexpr int blk = decltype(bc)::value;  // 0: (0,0)  1: (1,0)  2: (1,1)
      constexpr int ta = blk >= 1, tb = blk == 2;
      d4 acc = mf::zero4();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = mf::mfma(fr[ta][ks], fr[tb][ks], acc);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int a = 16 * ta + 4 * v + G.g, b = 16 * tb + G.j;
        if (a < d && b < d) {
          const double w = acc[v] + (a == b ? m1 * m1 : 0.0);
          WM[a * W::LDd + b] = w;
          if (a == b) sm[W::WD + a] = w;  // diag(H Q H') for the error estimate (src/perform_step.jl:148-158)
        }
      }
    });
    if (new_table) {  // the coefficient tables change with the step size only
      for (int e = G.lane; e < NB * NB * NB; e += 64) {
        const int a = e / (NB * NB), b = (e / NB) % NB, k = e % NB;
        sm[W::CF1 + e] = pc.At[a][k] * (tabL[kTabPJ + k] * tabL[kTabPJ + b]);
      }
      for (int e = G.lane; e < NB * NB; e += 64) sm[W::CF2 + e] = pc.At[e / NB][e % NB];
    }
    tv::lds_sync();
  }
  // Hs0[k][a] = H0[a][k] in padded (tile) indices, zero elsewhere -- only once E = T H' of the running step is done
  ODEF_MF_FN void chain_c(double* __restrict__ sm, int lane) {
    const double* H0 = sm + W::H0;
    double* hs0 = sm + W::HS0;
    const int ap_ = lane & 31, a = (ap_ >> 4) * TR + (ap_ & 15);
    const bool a_ok = (ap_ & 15) < TR;
#pragma unroll 4
    for (int kp_ = lane >> 5; kp_ < 32; kp_ += 2) {
      const bool ok = a_ok && (kp_ & 15) < TR;
      hs0[kp_ * LDP + ap_] = ok ? H0[a * d + (kp_ >> 4) * TR + (kp_ & 15)] : 0.0;
    }
    if (lane < 32) {  // the two pitch columns
      hs0[lane * LDP + 32] = 0.0;
      hs0[lane * LDP + 33] = 0.0;
    }
    tv::lds_sync();
  }

  // ---------------------------------------------------------------------------------------------------------- step
  // Compiled twice from the same source (as filter_tiles.h): HELPER = the wavefront without tiles.  Both instantiations
  // execute the same sequence of barriers; neither carries the other's registers.
  template <bool HELPER>
  ODEF_MF_FN void step(const PriorConsts& pc, const double* __restrict__ p, const double* __restrict__ tab,
                       const double* __restrict__ tab_next /* null after the last step */, int fixed_diffusion, int success_iter,
                       double* __restrict__ sm, d4 (&T)[NS], const Slots& S, const Geo& G0, int tid, int wave) {
    Geo G = fresh(G0);
    double* ex = sm + W::EX;
    double* vp = sm + W::VP;
    double* kp = sm + W::KP;
    double* tl = sm + W::TL;
    double* WM = sm + W::WM;
    double* wl = sm + W::WL;
    double* hs0 = sm + W::HS0;
    double* m = sm + W::MV;
    double* mp = sm + W::MP;
    double* z = sm + W::Z;
    double* y = sm + W::YV;
    double* sc = sm + W::SC;
    const double h1 = tab[kTabPIJ + 1];
    (void)m; (void)mp; (void)z; (void)y; (void)WM; (void)wl; (void)kp; (void)tl; (void)ex; (void)p; (void)success_iter; (void)tab_next; (void)wave; (void)S;

    ODEF_MF_STAMP(0)
    // x~ = P x (src/perform_step.jl:36-38): the covariance tiles go to the exchange as they are, the scaling is folded
    // into the coefficients of the congruence.  Mean, measurement and H Q H' of this step are already there (chain_*).
    if constexpr (!HELPER) ex_put_all(T, S, G0, ex);
    __syncthreads();
    ODEF_MF_STAMP(3)
    if constexpr (HELPER) {
      // sigma^2 = z' W^-1 z / d (src/diffusions.jl:72-80): blocked Cholesky of W = H Q H' beside the congruence of the
      // others -- first half here, second half beside stage 2
      if (!fixed_diffusion) {
        double* c11 = sm + W::CW11;
        double* c22 = sm + W::CW22;
        double* c21 = sm + W::CW21;
        for (int e = G.lane; e < 256; e += 64) {
          const int r = e >> 4, c = e & 15;
          c11[e] = (d >= 16 || (r < d && c < d)) ? WM[r * W::LDd + c] : 0.0;  // (d < 16: nothing of W lies beyond row / column d)
          c21[e] = (16 + r < d) ? WM[(16 + r) * W::LDd + c] : 0.0;
          c22[e] = (16 + r < d && 16 + c < d) ? WM[(16 + r) * W::LDd + 16 + c] : 0.0;
        }
        tv::lds_sync();
        ODEF_MF_HSTAMP(0)
        chol2_a(c11, c22, c21, sm + W::CWL, sm + W::CWW1, 16, G);
        ODEF_MF_HSTAMP(1)
      }
    } else {
      stage1(T, S, G0, ex, sm + W::CF1);
    }
    __syncthreads();
    ODEF_MF_STAMP(4)
    if constexpr (!HELPER) ex_put_all(T, S, G0, ex);
    __syncthreads();
    ODEF_MF_STAMP(5)
    G = fresh(G0);
    if constexpr (HELPER) {
      if (!fixed_diffusion) {
        double* w1 = sm + W::CWW1;
        ODEF_MF_HSTAMP(0)
        // y = L^-1 z: y1 = W11 z1 (the inverse of the first block exists: L21 = S21 W11' needed it), y2 from L22 y2 = z2 - L21 y1
        // by substitution inside the factorisation of the second block (its inverse is needed by nobody)
        const int l = G.lane, r = l & 15;
        double y1 = 0.0;
#pragma unroll
        for (int b = 0; b < 16; ++b) y1 += w1[r * 16 + b] * z[b];  // the inverses are lower triangular: fixed trip counts
        double t2 = z[16 + r];  // zero beyond d
#pragma unroll
        for (int b = 0; b < 16; ++b) t2 -= sm[W::CWL + r * 16 + b] * __shfl(y1, b, 64);
        ODEF_MF_HSTAMP(2)
        const double y2 = mf::diag_block_factor_solve(sm + W::CW22, t2);
        const double acc = wave_sum(l < 16 ? y1 * y1 : l < 32 ? y2 * y2 : 0.0);
        if (l == 0) {
          sc[0] = acc / d;
          sc[4] = acc / d;
        }
        ODEF_MF_HSTAMP(3)
      }
    } else {
      stage2(T, S, G0, ex, sm + W::CF2);
    }
    __syncthreads();  // the exchange is dead: region R0 now holds the panels
    ODEF_MF_STAMP(6)
    if constexpr (!HELPER) tl_put(T, S, G0, tl, true);
    __syncthreads();
    ODEF_MF_STAMP(7)
    if constexpr (!HELPER) {
      diag_resym(T, S, G0, tl);
      hproject(T, S, G0, hs0, tl, h1, vp);  // C0 = (A S A') H' into the V panel
    }
    __syncthreads();  // ... and sigma^2 is there
    ODEF_MF_STAMP(8)
    G = fresh(G0);
    const double sigma2_pred = fixed_diffusion ? 1.0 : sc[0];
    if constexpr (HELPER) {
      ODEF_MF_HSTAMP(0)
      // Sm = H C = H C0 + sigma2 H Q H' (d x d, plain index): the three blocks of its lower triangle, then its Cholesky
      // and W = L^-1 (src/perform_step.jl:66, src/filtering.jl:84-85)
      // (the operand fragments of the two row sets once: 32 loads instead of 48 in the helper's private instruction stream)
      double fa[2][8], fb[2][8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int a_ = 16 * t + G.j;                                    // A operand: row a of H (plain) = column pad_d(a) of Hs0
        const int acol = a_ < d ? pad_d(a_) : TR;                       // a zero column of Hs0 for the padding rows
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          fa[t][ks] = hs0[(4 * ks + G.g) * LDP + acol];
          fb[t][ks] = vp[(4 * ks + G.g) * LDP + G.j + 16 * t];
        }
      }
      static_for<0, 3>([&](auto bc) {
        constexpr int blk = decltype(bc)::value;  // 0: (0,0)  1: (1,0)  2: (1,1)
        constexpr int ta = blk >= 1, tb = blk == 2;
        d4 acc = mf::zero4();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = mf::mfma(fa[ta][ks], fb[tb][ks], acc);
        double* dst = sm + (blk == 0 ? W::SB11 : blk == 1 ? W::SB21 : W::SB22);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int a = 16 * ta + 4 * v + G.g, b = 16 * tb + G.j;
          double x = 0.0;
          if (a < d && b < d) x = acc[v] + h1 * vp[(32 + pad_d(a)) * LDP + b] + sigma2_pred * WM[a * W::LDd + b];
          dst[(4 * v + G.g) * 16 + G.j] = x;
        }
      });
      tv::lds_sync();
      ODEF_MF_HSTAMP(4)
      factor_s(sm + W::SB11, sm + W::SB22, sm + W::SB21, sm + W::L21, wl, G);
      ODEF_MF_HSTAMP(5)
    } else {
      add_sigma2_q(T, S, G0, pc, sigma2_pred);
    }
    __syncthreads();
    ODEF_MF_STAMP(10)
    G = fresh(G0);
    if constexpr (HELPER) {  // y = W z, z'Sm^-1 z, log det Sm: beside the V / K products of the others (y is read after the next barrier)
      ODEF_MF_HSTAMP(0)
      const int l = G.lane;
      double yv = 0.0, lg = 0.0;
      if (l < 32) {
#pragma unroll
        for (int b = 0; b < 32; ++b) yv += wl[l * LDP + b] * z[b];  // W_S is lower triangular, z is zero beyond d: fixed trip count
        const double rp = (l < 16) ? sm[W::SB11 + 256 + l] : sm[W::SB22 + 256 + l - 16];
        lg = (rp > 0.0) ? -log(rp) : 0.0;
        y[l] = yv;
      }
      const double zSz = wave_sum(yv * yv), logacc = wave_sum(lg);
      if (l == 0) {
        sc[1] = zSz;
        sc[3] = sc[3] - 0.5 * (zSz + 2.0 * logacc + d * 1.8378770664093453);
        if (fixed_diffusion) {  // src/diffusions.jl:11-36
          const double dt_ = zSz / d;
          const double prev = sc[4];
          sc[0] = dt_;
          sc[4] = static_diffusion_update<d>(fixed_diffusion, success_iter, prev, dt_);
        }
      }
      ODEF_MF_HSTAMP(6)
    } else {
      // per tile row R of the panels, one wavefront: C = C0 + sigma2 Q H', then V = C W' (in place) and K = V W
      for (int R = wave; R < NT; R += kMfTileWaves) {
        {
          const int bq = R >> 1, hr = R & 1;
          const double q0 = sigma2_pred * pc.Qt[bq][0], q1 = sigma2_pred * pc.Qt[bq][1] * h1;
          // (Q H')[r][a] = Qt[b][0] H0[a][i] + Qt[b][1] h1 [a == i],  r = (b, i): lane -> row ii = lane / 4, columns lane % 4 + 4 t
          const int ii = G.lane >> 2, a0 = G.lane & 3;
          if (ii < TR) {
            const int i = TR * hr + ii;
            double* row = vp + (16 * R + ii) * LDP;
            const double* hrow = hs0 + pad_d(i) * LDP;
#pragma unroll
            for (int t = 0; t < KS; ++t) {
              const int a = a0 + 4 * t;
              if (a < d) row[a] += q0 * hrow[a < TR ? a : a + 16 - TR] + (a == i ? q1 : 0.0);
            }
          }
        }
        tv::lds_sync();
        // W_S is lower triangular: the k-steps that meet only its zero blocks are skipped (21 products instead of 28)
        d4 v0 = mf::zero4(), v1 = mf::zero4();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const double a = vp[(16 * R + G.j) * LDP + 4 * ks + G.g];
          if (ks < 4) v0 = mf::mfma(a, wl[G.j * LDP + 4 * ks + G.g], v0);  // V[:, j < 16] = sum_{k <= j} C[:, k] W[j][k]
          v1 = mf::mfma(a, wl[(16 + G.j) * LDP + 4 * ks + G.g], v1);
        }
        tv::lds_sync();
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          vp[(16 * R + 4 * v + G.g) * LDP + G.j] = v0[v];
          vp[(16 * R + 4 * v + G.g) * LDP + 16 + G.j] = v1[v];
        }
        tv::lds_sync();
        d4 k0 = mf::zero4(), k1 = mf::zero4();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const double a = vp[(16 * R + G.j) * LDP + 4 * ks + G.g];
          k0 = mf::mfma(a, wl[(4 * ks + G.g) * LDP + G.j], k0);
          if (ks >= 4) k1 = mf::mfma(a, wl[(4 * ks + G.g) * LDP + 16 + G.j], k1);  // K[:, j >= 16] = sum_{k >= j} V[:, k] W[k][j]
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          kp[(16 * R + 4 * v + G.g) * LDP + G.j] = k0[v];
          kp[(16 * R + 4 * v + G.g) * LDP + 16 + G.j] = k1[v];
        }
      }
    }
    __syncthreads();
    ODEF_MF_STAMP(11)
    // m = m^- - V y (src/filtering.jl:87), un-preconditioned (src/perform_step.jl:75);  T = S^- - V V'
    if (tid < D) {
      const int prow = 32 * (tid / d) + pad_d(tid % d);
      double s = mp[tid];
#pragma unroll 7
      for (int a = 0; a < d; ++a) s -= vp[prow * LDP + a] * y[a];
      m[tid] = tab[kTabPIJ + tid / d] * s;
    }
    if constexpr (!HELPER) {
      rank_update(T, S, G0, vp, vp, nullptr);  // bitwise symmetric on the diagonal tiles (same products, same order)
      tl_put(T, S, G0, tl, false);                     // ... and the copies of the first-column tiles for E = T H'
    }
    __syncthreads();
    ODEF_MF_STAMP(12)
    if constexpr (HELPER) {
      ODEF_MF_HSTAMP(0)
      if (tab_next) chain_a1(pc, p, tab_next, tab_next != tab, sm, G0.lane);  // the next step's measurement chain beside the rest of this step
      ODEF_MF_HSTAMP(7)
    } else {
      hproject(T, S, G0, hs0, tl, h1, vp);  // E = T H' over the V panel
    }
    __syncthreads();
    ODEF_MF_STAMP(14)
    if constexpr (HELPER) {
      if (tab_next) {
        ODEF_MF_HSTAMP(0)
        chain_a2(pc, sm, G0.lane);
        ODEF_MF_HSTAMP(8)
        chain_b(pc, tab_next != tab, sm, fresh(G0));
        ODEF_MF_HSTAMP(9)
        chain_c(sm, G0.lane);
        ODEF_MF_HSTAMP(10)
      }
    } else {
      rank_update(T, S, G0, vp, kp, tab + kTabPIPI);  // S = T - E K', un-preconditioned (src/perform_step.jl:73-75)
    }
    __syncthreads();
    ODEF_MF_STAMP(15)
  }

  // one saved record: mean, packed covariance (own tiles), diffusion
  template <bool HELPER>
  ODEF_MF_FN void save_record(const FilterParams& P, long i, long slot, double diffusion, const double* __restrict__ sm,
                              const d4 (&T)[NS], const Slots& S, const Geo& G, int tid) {
    const double* m = sm + W::MV;
    const size_t N = (size_t)P.N;
    if (tid < D) P.mean[((size_t)slot * D + tid) * N + i] = m[tid];
    if constexpr (!HELPER) {
      // in place the elements of a record lie N doubles apart (8 useful bytes per line written); staged, the record is one
      // contiguous run that this workgroup completes line by line within the save
      const bool staged = P.cov_stage != nullptr;
      if (staged) {
        // element (a, b), a <= b, of tile (Q, Pc) sits at b (b + 1) / 2 + a of the record: with b = Pc TR + j, a = Q TR + 4 v + g that is
        // [a wavefront-uniform part] + (Pc TR) j + [j (j + 1) / 2 + g + 4 v] -- one 32-bit multiply-add per tile and lane (written as
        // 64-bit index arithmetic per element the save cost 3 ms of the every-step filter's 34 at 2 048 x 64: the step body
        // fills the instruction cache as it is).  Tried and dropped: the record written from the exchange copy of the tiles
        // at the start of the next step, by all nine wavefronts in one rolled, fully coalesced loop -- slower (39.0 against 37.2 ms).
        double* rec = P.cov_stage + ((size_t)slot * N + (size_t)i) * (size_t)P.stage_ld;
        const int lane_off = G.j * (G.j + 1) / 2 + G.g;
        static_for<0, NS>([&](auto sc_) {
          constexpr int s = decltype(sc_)::value;
          if (s < S.n) {
            const int Q = S.tq[s], Pc = S.tp[s], mcol = Pc * TR;
            const int base = mcol * (mcol + 1) / 2 + Q * TR;
            const int off = base + mcol * G.j + lane_off;
#pragma unroll
            for (int v = 0; v < 4; ++v)
              if (G.ok[v] && (Q < Pc || 4 * v + G.g <= G.j)) rec[off + 4 * v] = T[s][v];
          }
        });
        if (tid == 0) P.diff[(size_t)slot * N + i] = diffusion;
        return;
      }
      double* rec = P.cov + (size_t)slot * TRI * N + i;
      const size_t es = N;
      static_for<0, NS>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        if (s < S.n) {
          const int Q = S.tq[s], Pc = S.tp[s];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int a = Q * TR + 4 * v + G.g, b = Pc * TR + G.j;  // row <= column in the kept triangle
            if (G.ok[v] && a <= b) rec[(size_t)tri(b, a) * es] = T[s][v];
          }
        }
      });
    }
    if (tid == 0) P.diff[(size_t)slot * N + i] = diffusion;
  }

  // Adaptive solve of trajectory i on the same step (src/perform_step.jl:27-93 with OrdinaryDiffEq's PI controller, as
  // TilesFilter::run_adaptive does on the vector units): per attempt the controller (thread 0) fixes h and fills the
  // preconditioner table in LDS, the helper runs the measurement chain for it (not overlapped with the previous step: h is
  // only known now), all run `step`, the controller accepts or rejects.  A rejected attempt goes back to the previous
  // RECORD (every attempt is saved, §3.2): the mean and the covariance tiles are re-read from it -- the step has
  // overwritten both in place, and there is no room on chip for a second copy of 78 tiles.
  template <bool HELPER>
  ODEF_MF_FN void run_adaptive(const FilterParams& P, long i, int tid, double* __restrict__ sm) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (HELPER) __builtin_amdgcn_s_setprio(3);  // the helper's serial work is what the others wait for: first pick at issue and instruction fetch
    double* m = sm + W::MV;
    double* sc = sm + W::SC;
    const size_t N = (size_t)P.N;
    Geo G;
    G.lane = tid & 63;
    G.g = G.lane >> 4;
    G.j = tid & 15;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int ii = 4 * v + G.g;
      G.ok[v] = (ii < TR) && (G.j < TR);
      G.sym[v] = (ii < G.j ? ii : G.j) * TR + (ii < G.j ? G.j : ii);
    }
    Slots S;
    S.n = 0;
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      int qq = 0, pp = 0;
      static_for<0, kMfTileWaves>([&](auto wc) {
        constexpr int w = decltype(wc)::value;
        constexpr int cq = make_mf_own<NT>().Q[w][s], cp = make_mf_own<NT>().P[w][s], cn = make_mf_own<NT>().n[w];
        if (!HELPER && wave == w) {
          qq = cq < 0 ? 0 : cq;
          pp = cp < 0 ? 0 : cp;
          S.n = cn;
        }
      });
      S.tq[s] = qq;
      S.tp[s] = pp;
    });
    d4 T[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) T[s] = mf::zero4();
    __attribute__((unused)) double pl_local[RHS::np > 0 ? RHS::np : 1];
    const double* pl = pl_local;
    for (int k = 0; k < RHS::np; ++k) pl_local[k] = P.p_shared ? P.p[k] : P.p[(size_t)k * N + i];
    if constexpr (!HELPER) {
      if (tid == 0) {  // Taylor-mode initial mean (src/state_initialization.jl), zero covariance
        double u0[d], m0[D];
        for (int a = 0; a < d; ++a) u0[a] = P.u0[(size_t)a * N + i];
        taylor_init<RHS, q>(u0, pl, m0);
        for (int k = 0; k < D; ++k) m[k] = m0[k];
        for (int k = 0; k < 8; ++k) sc[k] = 0.0;
        for (int a = 0; a < d; ++a) sm[W::UC + a] = u0[a];
      }
      if (tid < 32) sm[W::Z + tid] = 0.0;  // the entries behind z[d-1] stay zero
    }
    __syncthreads();
    double* tabL = sm + W::TAB;
    double* wd = sm + W::WD;
    double* ucur = sm + W::UC;
    double* ctl = sm + W::CTL;  // [0] go on, [1] restore the previous state, [2] slot of this attempt's record, [3] naccept, [4] state finite
    save_record<HELPER>(P, i, 0, 0.0, sm, T, S, G, tid);
    const bool boss = !HELPER && tid == 0;  // controller state: meaningful in tile thread 0 only
    const Controller& ct = P.ctrl;
    double t = P.t0, h = P.dt0, qold = ct.qoldinit, q11 = 1.0, sc3_prev = 0.0, sc4_prev = 0.0;
    int naccept = 0, nreject = 0, nsaved = 1, ret = 0;
    long attempts = 0;
    const long max_attempts = 20 * P.max_save + 1000;
    if (boss) P.tsave[i] = P.t0;
    for (;;) {
      if (boss) {
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
      __syncthreads();
      if (ctl[0] == 0.0) break;  // workgroup-uniform
      if constexpr (HELPER) {
        chain_a1(P.pc, pl, tabL, true, sm, G.lane);  // (the table is in place: the copy inside is onto itself)
        chain_a2(P.pc, sm, G.lane);
        chain_b(P.pc, true, sm, fresh(G));
        chain_c(sm, G.lane);
      }
      __syncthreads();
      step<HELPER>(P.pc, pl, tabL, nullptr, P.fixed_diffusion, (int)ctl[3], sm, T, S, G, tid, wave);
      if (boss) {
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
      __syncthreads();
      const long slot = (long)ctl[2];
      if (ctl[1] != 0.0) {  // x_filt is not committed: back to the previous record, cache.x = P^-1 (P x) (:73)
        if (tid < D) {
          const double v = P.mean[((size_t)(slot - 1) * D + tid) * N + i];
          m[tid] = tabL[kTabPIJ + tid / d] * (tabL[kTabPJ + tid / d] * v);
        }
        if constexpr (!HELPER) {
          const double* rec = P.cov + (size_t)(slot - 1) * TRI * N + i;
          static_for<0, NS>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            if (s < S.n) {
              const int Q = S.tq[s], Pc = S.tp[s];
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const int a = Q * TR + 4 * v + G.g, b = Pc * TR + G.j;
                const int hi = a > b ? a : b, lo = a > b ? b : a;  // a diagonal tile holds both halves
                T[s][v] = G.ok[v] ? rec[(size_t)tri(hi, lo) * N] : 0.0;
              }
            }
          });
        }
      }
      save_record<HELPER>(P, i, slot, sc[4], sm, T, S, G, tid);
      if (boss) {
        ++nsaved;
        bool finite = true;
        for (int k = 0; k < D; ++k) finite = finite && (fabs(m[k]) <= 1.79769313486231570815e+308);
        if (!finite) ret = 3;
        ctl[4] = finite ? 1.0 : 0.0;  // its own word: ctl[0] is rewritten by thread 0 right after the next barrier
      }
      __syncthreads();
      if (ctl[4] == 0.0) break;
    }
    if (boss) {
      P.loglik[i] = sc[3];
      P.naccept[i] = naccept;
      P.nreject[i] = nreject;
      P.nf[i] = naccept + nreject;
      P.njac[i] = IS_EK1 ? naccept + nreject : 0;
      P.nsaved[i] = nsaved;
      P.retcode[i] = ret;
    }
  }

  // whole fixed-step solve of trajectory i
  template <bool HELPER>
  ODEF_MF_FN void run(const FilterParams& P, long i, int tid, double* __restrict__ sm) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef ODEF_MF_DEBUG_FILL  // (diagnostic: LDS pre-filled, NaN in [ODEF_MF_DEBUG_LO, ODEF_MF_DEBUG_HI) -- finds reads of LDS nobody wrote)
    for (int e = tid; e < W::size; e += kMfBlock) sm[e] = (e >= (ODEF_MF_DEBUG_LO) && e < (ODEF_MF_DEBUG_HI)) ? __builtin_nan("") : 0.0;
    __syncthreads();
#endif
    if constexpr (HELPER) __builtin_amdgcn_s_setprio(3);  // the helper's serial work is what the others wait for: first pick at issue and instruction fetch
    double* m = sm + W::MV;
    double* sc = sm + W::SC;
    const size_t N = (size_t)P.N;
    Geo G;
    G.lane = tid & 63;
    G.g = G.lane >> 4;
    G.j = tid & 15;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int ii = 4 * v + G.g;
      G.ok[v] = (ii < TR) && (G.j < TR);
      G.sym[v] = (ii < G.j ? ii : G.j) * TR + (ii < G.j ? G.j : ii);
    }
    Slots S;
    S.n = 0;
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      int qq = 0, pp = 0;
      static_for<0, kMfTileWaves>([&](auto wc) {
        constexpr int w = decltype(wc)::value;
        constexpr int cq = make_mf_own<NT>().Q[w][s], cp = make_mf_own<NT>().P[w][s], cn = make_mf_own<NT>().n[w];
        if (!HELPER && wave == w) {
          qq = cq < 0 ? 0 : cq;
          pp = cp < 0 ? 0 : cp;
          S.n = cn;
        }
      });
      S.tq[s] = qq;
      S.tp[s] = pp;
    });
    d4 T[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) T[s] = mf::zero4();
    __attribute__((unused)) double pl_local[RHS::np > 0 ? RHS::np : 1];
    const double* pl = pl_local;
    for (int k = 0; k < RHS::np; ++k) pl_local[k] = P.p_shared ? P.p[k] : P.p[(size_t)k * N + i];
    if constexpr (!HELPER) {
      if (tid == 0) {  // Taylor-mode initial mean (src/state_initialization.jl), zero covariance
        double u0[d], m0[D];
        for (int a = 0; a < d; ++a) u0[a] = P.u0[(size_t)a * N + i];
        taylor_init<RHS, q>(u0, pl, m0);
        for (int k = 0; k < D; ++k) m[k] = m0[k];
        for (int k = 0; k < 8; ++k) sc[k] = 0.0;
        for (int a = 0; a < d; ++a) sm[W::UC + a] = u0[a];
      }
      if (tid < 32) sm[W::Z + tid] = 0.0;  // the entries behind z[d-1] stay zero
    }
    __syncthreads();
    if (P.everystep) save_record<HELPER>(P, i, 0, 0.0, sm, T, S, G, tid);
    if constexpr (HELPER) {
      if (P.nsteps > 0) {
        const double* tab0 = P.ptab + (size_t)uniform_load(P.tab_idx) * kTabStride;
        chain_a1(P.pc, pl, tab0, true, sm, G.lane);
        chain_a2(P.pc, sm, G.lane);
        chain_b(P.pc, true, sm, fresh(G));
        chain_c(sm, G.lane);
      }
    }
    __syncthreads();
    for (long n = 0; n < P.nsteps; ++n) {
      const double* tab = P.ptab + (size_t)uniform_load(P.tab_idx + n) * kTabStride;
      const double* tab_next = n + 1 < P.nsteps ? P.ptab + (size_t)uniform_load(P.tab_idx + n + 1) * kTabStride : nullptr;
      step<HELPER>(P.pc, pl, tab, tab_next, P.fixed_diffusion, (int)n, sm, T, S, G, tid, wave);
      if (P.everystep) save_record<HELPER>(P, i, n + 1, sc[4], sm, T, S, G, tid);
    }
    if (!P.everystep) save_record<HELPER>(P, i, 0, sc[4], sm, T, S, G, tid);
    if constexpr (!HELPER) {
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
  }
#undef ODEF_MF_FN
};

template <class RHS, int q, bool EK1>
__global__ __launch_bounds__(kMfBlock) void ek_filter_mfma_kernel(const FilterParams P) {
  using MF = MfmaFilter<RHS, q, EK1>;
  __shared__ double sm[MF::W::size];
  const long i = team_traj(P.N);  // XCD-aware: neighbouring trajectories share an L2
  if (i < 0) return;
  if (threadIdx.x >= 64 * kMfHelper)  // the helper wavefront: same barriers, its own code path and register allocation
    MF::template run<true>(P, i, (int)threadIdx.x, sm);
  else
    MF::template run<false>(P, i, (int)threadIdx.x, sm);
}

template <class RHS, int q, bool EK1>
__global__ __launch_bounds__(kMfBlock) void ek_filter_mfma_adaptive_kernel(const FilterParams P) {
  using MF = MfmaFilter<RHS, q, EK1>;
  __shared__ double sm[MF::W::size];
  const long i = team_traj(P.N);
  if (i < 0) return;
  if (threadIdx.x >= 64 * kMfHelper)
    MF::template run_adaptive<true>(P, i, (int)threadIdx.x, sm);
  else
    MF::template run_adaptive<false>(P, i, (int)threadIdx.x, sm);
}

}  // namespace odef
#endif
