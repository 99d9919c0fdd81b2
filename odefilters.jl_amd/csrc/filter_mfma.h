// EK0/EK1 fixed-step filter for large state dimension (Pleiades: d = 28, D = 168) on the FP64 matrix cores: one
// 512-thread workgroup per trajectory, the covariance DISTRIBUTED IN REGISTERS as 16 x 16 accumulator tiles of
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
// factorisation, no Householder QR and no per-pivot synchronisation of the workgroup: the step has 16 barriers instead
// of ~45 (filter_tiles.h, which stays in the library: adaptive solves, ODEF_PLEIADES_FILTER=tiles).
//
// Tiles.  Each derivative block (d = 28 rows) is split into two tiles of 14 real rows + 2 zero rows, so that the
// Kronecker congruence maps whole tiles onto whole tiles (tile 2 b + h <-> rows 14 h .. 14 h + 13 of block b) and
// DP = 32 (q + 1) = 192 at order 5.  Only tiles (Q, P) with Q <= P are kept (78 of them): tile column P belongs to ONE
// wavefront (columns are dealt to 7 wavefronts by decreasing size: 12, 11, 11, 11, 11, 11, 11 tiles), in the accumulator layout
//     register v of lane l  <->  element (row 4 v + l / 16, column l % 16).
// A tile in this layout IS the B operand of a K = 16 product (register v = k-step v), so  H (.) tile  needs no data
// movement; products that need the tile as the A operand (the six tiles above the diagonal of the first two derivative
// blocks) go through a small LDS copy.  The rank-d updates read their D x d operand panels (V, K, E) from LDS.
// The eighth wavefront owns no tile: it runs the two d x d factorisations (H Q H' for sigma^2 -- beside the congruence
// of the others: sigma^2 Q is added to the tiles and sigma^2 Q H' to the panel only after C = (A S A') H' is there -- and
// Sm, which everybody waits for).
#pragma once
#ifndef ODEF_HOST_EMUL
#include "ek_lane.h"
#include "mfma_dense.h"
#include "team.h"
#include "wave_vec.h"

namespace odef {

constexpr int kMfWaves = 9;       // wavefronts of the workgroup
constexpr int kMfTileWaves = 8;   // ... that own tiles; the last one is the helper
constexpr int kMfHelper = 8;
constexpr int kMfBlock = 64 * kMfWaves;

// In-kernel stamps at phase boundaries: ONLY in the diagnostic build of tools/mfma_filter_stamps.hip
// (-DODEF_MF_STAMPS); the product kernel contains none.
#ifdef ODEF_MF_STAMPS
__device__ unsigned long long* g_mf_stamp_buf = nullptr;
__device__ unsigned long long g_mf_stamp_last = 0;
#define ODEF_MF_STAMP(k)                                                      \
  if (tid == 0 && blockIdx.x == 0 && g_mf_stamp_buf) {                        \
    unsigned long long t_;                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    g_mf_stamp_buf[(k)] += t_ - g_mf_stamp_last;                              \
    g_mf_stamp_last = t_;                                                     \
  }
#else
#define ODEF_MF_STAMP(k)
#endif

// Tile ownership.  The "head" of a tile column (its tiles in the first four tile rows: the ones H (.) reads) stays in
// one wavefront, so that C = S^- H' needs no reduction across wavefronts; every other tile may go anywhere.  Heads are
// dealt first (largest column first, to the least loaded wavefront), then the remaining tiles column by column, each to
// the least loaded wavefront (staying with the previous tile's wavefront while that one is below the average load, so
// that the B fragments of a column are reloaded rarely).  78 tiles over 8 wavefronts: 10, 10, 10, 10, 10, 10, 9, 9.
template <int NT>
struct MfOwnTab {
  static constexpr int ntiles = NT * (NT + 1) / 2;
  static constexpr int cap = (ntiles + kMfTileWaves - 1) / kMfTileWaves + 3;  // upper bound of a wavefront's load
  int n[kMfTileWaves];
  int Q[kMfTileWaves][cap];
  int P[kMfTileWaves][cap];
  int maxload;
  bool ok;
};
template <int NT>
constexpr MfOwnTab<NT> make_mf_own() {
  MfOwnTab<NT> t{};
  constexpr int cap = MfOwnTab<NT>::cap, target = (MfOwnTab<NT>::ntiles + kMfTileWaves - 1) / kMfTileWaves;
  t.ok = true;
  for (int w = 0; w < kMfTileWaves; ++w) {
    t.n[w] = 0;
    for (int s = 0; s < cap; ++s) {
      t.Q[w][s] = -1;
      t.P[w][s] = -1;
    }
  }
  auto least = [&]() {
    int w = 0;
    for (int k = 1; k < kMfTileWaves; ++k)
      if (t.n[k] < t.n[w]) w = k;
    return w;
  };
  for (int P = NT - 1; P >= 0; --P) {  // heads
    const int w = least();
    for (int Q = 0; Q <= (P < 3 ? P : 3); ++Q) {
      if (t.n[w] >= cap) { t.ok = false; return t; }
      t.Q[w][t.n[w]] = Q;
      t.P[w][t.n[w]] = P;
      ++t.n[w];
    }
  }
  int prev = -1;
  for (int P = NT - 1; P >= 4; --P)
    for (int Q = 4; Q <= P; ++Q) {
      int w = least();
      if (prev >= 0 && t.n[prev] < target) w = prev;
      if (t.n[w] >= cap) { t.ok = false; return t; }
      t.Q[w][t.n[w]] = Q;
      t.P[w][t.n[w]] = P;
      ++t.n[w];
      prev = w;
    }
  t.maxload = 0;
  for (int w = 0; w < kMfTileWaves; ++w)
    if (t.n[w] > t.maxload) t.maxload = t.n[w];
  return t;
}

template <int d, int NB>
struct MfLds {
  static constexpr int D = d * NB, NT = 2 * NB, DP = 16 * NT, ntiles = NT * (NT + 1) / 2;
  static constexpr int TR = d / 2, TSZ = TR * TR;  // real rows per tile; one tile in the exchange
  static constexpr int LDP = 34;                   // pitch of the operand panels (32 + 2: conflict-light fragment reads)
  // region R0, time-shared: tile exchange of the congruence  |  panels V / E, K + the transposed-role tiles + scratch
  static constexpr int EX = 0, EX_size = ntiles * TSZ;
  static constexpr int VP = 0, KP = VP + DP * LDP, TL = KP + DP * LDP;  // TL: 6 tiles above the diagonal + 4 diagonal ones
  static constexpr int R0_need = TL + 10 * 256;
  static constexpr int R0_size = EX_size > R0_need ? EX_size : R0_need;
  // region B, time-shared:
  //   H0 | M0 | WM          raw / scaled Jacobian block, M0, W = H Q H'           (measure ... chol(W))
  //   scratch of chol(W)    over H0 | M0 once W is built: two diagonal blocks (+ pivots), the block below, L21, W11, W22
  //   WL | scratch of Sm    W_S = L^-1 of the innovation covariance [32][LDP], then the blocks of Sm and L21
  static constexpr int LDd = d + 1;
  static constexpr int H0 = R0_size, M0 = H0 + d * d, WM = M0 + d * d;
  static constexpr int CW11 = R0_size, CW22 = CW11 + 272, CW21 = CW22 + 272, CWL = CW21 + 256, CWW1 = CWL + 256, CWW2 = CWW1 + 256;
  static_assert(CWW2 + 256 <= WM, "the scratch of chol(W) must not reach W itself");
  static constexpr int WL = R0_size, SB11 = WL + 32 * LDP, SB22 = SB11 + 272, SB21 = SB22 + 272, L21 = SB21 + 256;
  static constexpr int B_need1 = 2 * d * d + d * LDd, B_need2 = L21 + 256 - R0_size;
  static constexpr int B_size = B_need1 > B_need2 ? B_need1 : B_need2;
  static constexpr int HS0 = R0_size + B_size;  // [32][LDP]: H0' in padded indices (k = state column, a = measurement)
  static constexpr int MV = HS0 + 32 * LDP, MT = MV + D, MP = MT + D, Z = MP + D, YV = Z + 32, UP = YV + 32;  // z: d values + zeros up to 32
  static constexpr int DU = UP + d, WD = DU + d, UC = WD + d, SC = UC + d, CTL = SC + 8, TAB = CTL + 8;
  static constexpr int CF1 = TAB + kTabStride, CF2 = CF1 + NB * NB * NB;  // coefficients of the two congruence stages
  static constexpr int size = CF2 + NB * NB;
  static_assert(d % 2 == 0 && d / 2 <= 16 && d / 2 >= 1, "a derivative block is split into two tiles of d / 2 <= 16 rows");
  static constexpr int DUMMY = EX_size;  // 64 doubles behind the exchange: where the padding lanes of a tile store go
  static_assert(R0_size >= EX_size + 64, "room for the dummy stores");
  static_assert(size * 8 <= 160 * 1024, "LDS budget of one workgroup");
};

template <class RHS, int q, bool IS_EK1>
struct MfmaFilter {
  static constexpr int d = RHS::d, NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2;
  using W = MfLds<d, NB>;
  using d4 = mf::d4;
  static constexpr int NT = W::NT, TR = W::TR, TSZ = W::TSZ, LDP = W::LDP, NTHR = kMfBlock;
  static constexpr int NS = make_mf_own<NT>().maxload;  // tile slots per wavefront
  static constexpr int kHelper = kMfHelper;
  static_assert(make_mf_own<NT>().ok, "tile columns do not fit the slot table");

  struct Geo {  // per-lane geometry of the accumulator layout
    int g, j, lane;
    bool ok[4];  // element (4 v + g, j) is a real entry of its tile
    int sym[4];  // offset of element (min, max) of (4 v + g, j) in a compact TR x TR tile: upper-triangle read of a diagonal tile
  };
  struct Slots {
    int tq[NS], tp[NS];  // wave-uniform tile coordinates of slot s (-1: unused)
  };
  // The lane geometry is loop-invariant, and so is every LDS address derived from it: left alone, the compiler hoists a few
  // hundred of them out of the step loop, runs out of registers and reloads them from scratch inside the phases
  // (scratch loads share vmcnt with nothing else here, but they sit on the critical path of every phase).  Each phase
  // therefore starts from a laundered copy -- no instructions, the values just stop being provably invariant.
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

  ODEF_MF_FN int pad_d(int a) { return 16 * (a / TR) + a % TR; }  // measurement / in-block index -> padded
  ODEF_MF_FN int uidx(int Q, int P) { return P * (P + 1) / 2 + Q; }

  // tile <-> compact exchange slot
  ODEF_MF_FN void ex_put(double* __restrict__ ex, int u, const Geo& G, const d4& t) {
    const int base = u * TSZ + G.g * TR + G.j;
#pragma unroll
    for (int v = 0; v < 4; ++v) ex[G.ok[v] ? base + 4 * v * TR : W::DUMMY + G.lane] = t[v];  // no exec masking
  }

  // (C_P)' = Hs' (.) over the tiles of the first two derivative blocks of every tile column, panel OUT[row][a]:
  //   block 0 (tiles Q = 0, 1):  H0-part, 8 MFMAs per tile;  block 1 (tiles 2, 3):  h1 I, one FMA per element
  ODEF_MF_FN void hproject(const d4 (&T)[NS], const Slots& S, const Geo& G0, const double* __restrict__ hs0,
                           const double* __restrict__ tl, double h1, double* __restrict__ out) {
    const Geo G = fresh(G0);
    d4 acc0 = mf::zero4(), acc1 = mf::zero4();
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      const int Q = S.tq[s], P = S.tp[s];
      if (Q >= 0) {
        if (Q <= 1) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const double* hp = hs0 + (16 * Q + 4 * ks + G.g) * LDP + G.j;
            acc0 = mf::mfma(hp[0], T[s][ks], acc0);
            acc1 = mf::mfma(hp[16], T[s][ks], acc1);
          }
        } else if (Q == 2) {
          acc0 += h1 * T[s];
        } else if (Q == 3) {
          acc1 += h1 * T[s];
        }
        if (Q == (P < 3 ? P : 3)) {  // last head tile of the column: the sources ABOVE the diagonal come transposed from the LDS copies
#pragma unroll
          for (int Qc = 1; Qc <= 3; ++Qc) {
            if (Qc > P) {
              const double* tq = tl + (Qc * (Qc - 1) / 2 + P) * 256;
              if (Qc == 1) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                  const double* hp = hs0 + (16 * Qc + 4 * ks + G.g) * LDP + G.j;
                  const double b = tq[G.j * 16 + 4 * ks + G.g];
                  acc0 = mf::mfma(hp[0], b, acc0);
                  acc1 = mf::mfma(hp[16], b, acc1);
                }
              } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                  const double x = h1 * tq[G.j * 16 + 4 * v + G.g];
                  if (Qc == 2) acc0[v] += x; else acc1[v] += x;
                }
              }
            }
          }
          // the panel is indexed by the PLAIN measurement index a (28 real columns + 4 zero ones: 7 k-steps in the rank
          // updates instead of 8): row i of accumulator ta is a = 14 ta + i for i < 14; its two zero rows go to 28 + 2 ta + (i - 14)
          double* o = out + (16 * P + G.j) * LDP;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int i = 4 * v + G.g;
            o[i < TR ? i : 2 * TR + (i - TR)] = acc0[v];
            o[i < TR ? TR + i : 2 * TR + 2 + (i - TR)] = acc1[v];
          }
          acc0 = mf::zero4();
          acc1 = mf::zero4();
        }
      }
    });
  }

  // the ten tiles of the first four tile columns, as full 16 x 16 row-major copies: the six above the diagonal (read
  // back transposed by hproject) and the four diagonal ones (read back through their upper triangle by diag_resym)
  ODEF_MF_FN void tl_put(const d4 (&T)[NS], const Slots& S, const Geo& G0, double* __restrict__ tl, bool with_diag) {
    const Geo G = fresh(G0);
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      const int Q = S.tq[s], P = S.tp[s];
      if (Q >= 0 && P <= 3 && (Q < P || with_diag)) {
        double* dst = tl + (Q < P ? P * (P - 1) / 2 + Q : 6 + Q) * 256 + G.g * 16 + G.j;
#pragma unroll
        for (int v = 0; v < 4; ++v) dst[64 * v] = T[s][v];
      }
    });
  }
  // A diagonal tile must be EXACTLY symmetric where it enters a product as a whole (H (.) tile): its antisymmetric part
  // is not damped by the update but multiplied by (I + K H), step after step (numpy model in DESIGN.md section 3.9).
  ODEF_MF_FN void diag_resym(d4 (&T)[NS], const Slots& S, const Geo& G0, const double* __restrict__ tl) {
    const Geo G = fresh(G0);
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      const int Q = S.tq[s], P = S.tp[s];
      if (Q >= 0 && Q == P && P <= 3) {
        const double* src = tl + (6 + Q) * 256;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int i = 4 * v + G.g;
          T[s][v] = src[(i < G.j ? i : G.j) * 16 + (i < G.j ? G.j : i)];
        }
      }
    });
  }

  static constexpr int KS = (d + 3) / 4;  // k-steps of a product over the measurement index
  // T[s] -= A_panel[tile row Q] B_panel[tile row P]'  (rank-d update of every tile)
  ODEF_MF_FN void rank_update(d4 (&T)[NS], const Slots& S, const Geo& G0, const double* __restrict__ ap,
                              const double* __restrict__ bp) {
    const Geo G = fresh(G0);
    double bf[KS];
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      const int Q = S.tq[s], P = S.tp[s];
      if (Q >= 0) {
        bool reload = true;
        if constexpr (s > 0) reload = S.tp[s - 1] != P;
        if (reload) {  // first tile of a run of one column: its B fragments (negated: the product is subtracted)
          const double* b = bp + (16 * P + G.j) * LDP + G.g;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) bf[ks] = -b[4 * ks];
        }
        const double* a = ap + (16 * Q + G.j) * LDP + G.g;
        d4 acc = T[s];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mf::mfma(a[4 * ks], bf[ks], acc);
        T[s] = acc;
      }
    });
  }

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
  // sum over the wavefront
  ODEF_MF_FN double wave_sum(double x) {
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) x += __shfl_xor(x, msk, 64);
    return x;
  }

  // ---------------------------------------------------------------------------------------------------------- step
  // Compiled twice from the same source (as filter_tiles.h): HELPER = the wavefront without tiles.  Both instantiations
  // execute the same number of barriers; neither carries the other's registers.
  template <bool HELPER>
  ODEF_MF_FN void step(const PriorConsts& pc, const double* __restrict__ p, const double* __restrict__ tab, int fixed_diffusion,
                       int success_iter, double* __restrict__ sm, d4 (&T)[NS], const Slots& S, const Geo& G0, int tid, int wave) {
    Geo G = fresh(G0);
    double* ex = sm + W::EX;
    double* vp = sm + W::VP;
    double* kp = sm + W::KP;
    double* tl = sm + W::TL;
    double* H0 = sm + W::H0;
    double* M0 = sm + W::M0;
    double* WM = sm + W::WM;
    double* wl = sm + W::WL;
    double* hs0 = sm + W::HS0;
    double* m = sm + W::MV;
    double* mt = sm + W::MT;
    double* mp = sm + W::MP;
    double* z = sm + W::Z;
    double* y = sm + W::YV;
    double* up = sm + W::UP;
    double* du = sm + W::DU;
    double* sc = sm + W::SC;
    const double pi0 = tab[kTabPIJ + 0], pi1 = tab[kTabPIJ + 1], h1 = pi1;

    ODEF_MF_STAMP(0)
    // x~ = P x (src/perform_step.jl:36-38): the covariance tiles go to the exchange as they are, the scaling is folded
    // into the coefficients of the congruence
    if (tid < D) mt[tid] = tab[kTabPJ + tid / d] * m[tid];
    if constexpr (HELPER) {  // coefficient tables of the congruence (read as LDS broadcasts: no scalar loads inside the tile loops)
      for (int e = G.lane; e < NB * NB * NB; e += 64) {
        const int a = e / (NB * NB), b = (e / NB) % NB, k = e % NB;
        sm[W::CF1 + e] = pc.At[a][k] * (tab[kTabPJ + k] * tab[kTabPJ + b]);
      }
      for (int e = G.lane; e < NB * NB; e += 64) sm[W::CF2 + e] = pc.At[e / NB][e % NB];
    }
    if constexpr (!HELPER) {
      static_for<0, NS>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        if (S.tq[s] >= 0) ex_put(ex, uidx(S.tq[s], S.tp[s]), G, T[s]);
      });
    }
    __syncthreads();
    // m^- = A m~ , u_pred  (src/filtering.jl:22-25, src/perform_step.jl:43)
    if (tid < D) {
      const int J = tid / d, a = tid % d;
      double s = mt[tid];
      for (int j = J + 1; j < NB; ++j) s += pc.At[J][j] * mt[j * d + a];
      mp[tid] = s;
      if (tid < d) up[tid] = pi0 * s;
    }
    __syncthreads();
    ODEF_MF_STAMP(1)
    // measure! (src/perform_step.jl:95-132)
    if constexpr (HasTeamEval<RHS>::value) {
      static_assert(RHS::team_scratch <= d * W::LDd, "pair buffer must fit the W area");
      RHS::team_eval_pairs(tid, up, WM);
      __syncthreads();
      RHS::team_eval_assemble(tid, NTHR, up, WM, du, IS_EK1 ? H0 : nullptr);  // raw J into H0, scaled below
      __syncthreads();
    } else {
      if (tid == 0) {
        double u_[d], du_[d];
        for (int a = 0; a < d; ++a) u_[a] = up[a];
        RHS::f(u_, p, du_);
        for (int a = 0; a < d; ++a) du[a] = du_[a];
        if constexpr (IS_EK1) RHS::jac(u_, p, *reinterpret_cast<double (*)[d][d]>(H0));
      }
      __syncthreads();
    }
    ODEF_MF_STAMP(2)
    if (tid < d) z[tid] = pi1 * mp[d + tid] - du[tid];
    for (int e = tid; e < d * d; e += NTHR) {  // H0 = -J pi0 ; M0 = H0 QL00 + I h1 QL10 (src/diffusions.jl:78)
      const int r = e / d, a = e % d;
      double h0 = 0.0;
      if constexpr (IS_EK1) h0 = (0.0 - H0[e]) * pi0;
      H0[e] = h0;
      M0[e] = h0 * pc.QLt[0][0] + (r == a ? h1 * pc.QLt[1][0] : 0.0);
    }
    __syncthreads();
    {
      const double m1 = h1 * pc.QLt[1][1];
      for (int e = tid; e < d * d; e += NTHR) {
        const int r = e / d, s_ = e % d;
        double acc = (r == s_) ? m1 * m1 : 0.0;
        for (int a = 0; a < d; ++a) acc += M0[r * d + a] * M0[s_ * d + a];
        WM[r * W::LDd + s_] = acc;
        if (r == s_) sm[W::WD + r] = acc;  // diag(H Q H') for the error estimate (src/perform_step.jl:148-158)
      }
      for (int e = tid; e < 32 * LDP; e += NTHR) {  // Hs0[k][a] = H0[a][k] in padded (tile) indices, zero elsewhere
        const int kp_ = e / LDP, ap_ = e % LDP;
        double v = 0.0;
        if (ap_ < 32 && (kp_ & 15) < TR && (ap_ & 15) < TR) v = H0[((ap_ >> 4) * TR + (ap_ & 15)) * d + (kp_ >> 4) * TR + (kp_ & 15)];
        hs0[e] = v;
      }
    }
    __syncthreads();
    ODEF_MF_STAMP(3)
    G = fresh(G0);
    if constexpr (HELPER) {
      // sigma^2 = z' W^-1 z / d (src/diffusions.jl:72-80): blocked Cholesky of W = H Q H' beside the congruence of the
      // others -- first half here, second half beside stage 2
      if (!fixed_diffusion) {
        double* c11 = sm + W::CW11;
        double* c22 = sm + W::CW22;
        double* c21 = sm + W::CW21;
        for (int e = G.lane; e < 256; e += 64) {
          const int r = e >> 4, c = e & 15;
          c11[e] = WM[r * W::LDd + c];
          c21[e] = (16 + r < d) ? WM[(16 + r) * W::LDd + c] : 0.0;
          c22[e] = (16 + r < d && 16 + c < d) ? WM[(16 + r) * W::LDd + 16 + c] : 0.0;
        }
        tv::lds_sync();
        chol2_a(c11, c22, c21, sm + W::CWL, sm + W::CWW1, 16, G);
      }
    } else {
      // predict_cov! (src/filtering.jl:33-41) in two stages through the tile exchange, whole tiles onto whole tiles:
      //   Z(Q, P) = sum_{k >= Q/2} At[Q/2][k] pj[k] pj[P/2] S(2k + Q%2, P)     sources in the own tile COLUMN; those below the
      //                                                                        diagonal are read transposed (S symmetric)
      //   S^-(Q, P) = sum_{k >= P/2} At[P/2][k] Z(Q, 2k + P%2)  ( + sigma2 Qt later )   sources in the own tile ROW, all kept
      static_for<0, NS>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        const int Q = S.tq[s], P = S.tp[s];
        if (Q >= 0) {
          const int a = Q >> 1, hq = Q & 1, b = P >> 1;
          const double* cf = sm + W::CF1 + (a * NB + b) * NB;
          d4 acc = mf::zero4();
          for (int k = a; k < NB; ++k) {
            const int Qs = 2 * k + hq;
            const double coef = cf[k];
            if (Qs < P) {
              const double* src = ex + uidx(Qs, P) * TSZ + G.g * TR + G.j;
#pragma unroll
              for (int v = 0; v < 4; ++v) acc[v] += coef * src[4 * v * TR];
            } else if (Qs == P) {  // a diagonal tile is its upper triangle (the rank updates leave rounding-level asymmetry)
              const double* src = ex + uidx(Qs, P) * TSZ;
#pragma unroll
              for (int v = 0; v < 4; ++v) acc[v] += coef * src[G.sym[v]];
            } else {
              const double* src = ex + uidx(P, Qs) * TSZ + G.j * TR + G.g;
#pragma unroll
              for (int v = 0; v < 4; ++v) acc[v] += coef * src[4 * v];
            }
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) T[s][v] = G.ok[v] ? acc[v] : 0.0;
        }
      });
    }
    __syncthreads();
    ODEF_MF_STAMP(4)
    G = fresh(G0);
    if constexpr (!HELPER) {
      static_for<0, NS>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        if (S.tq[s] >= 0) ex_put(ex, uidx(S.tq[s], S.tp[s]), G, T[s]);
      });
    }
    __syncthreads();
    ODEF_MF_STAMP(5)
    G = fresh(G0);
    if constexpr (HELPER) {
      if (!fixed_diffusion) {
        double* w1 = sm + W::CWW1;
        double* w2 = sm + W::CWW2;
        chol2_b(sm + W::CW22, w2, 16);
        // y = L^-1 z: y1 = W11 z1, y2 = W22 (z2 - L21 y1); lanes 0..15 hold y1, lanes 16..31 hold y2
        const int l = G.lane, r = l & 15;
        double y1 = 0.0;
        for (int b = 0; b <= r; ++b) y1 += w1[r * 16 + b] * z[b];
        double t2 = z[16 + r];  // zero beyond d
        for (int b = 0; b < 16; ++b) t2 -= sm[W::CWL + r * 16 + b] * __shfl(y1, b, 64);
        double y2 = 0.0;
        for (int b = 0; b <= r; ++b) y2 += w2[r * 16 + b] * __shfl(t2, b, 64);
        const double acc = wave_sum(l < 16 ? y1 * y1 : l < 32 ? y2 * y2 : 0.0);
        if (l == 0) {
          sc[0] = acc / d;
          sc[4] = acc / d;
        }
      }
    } else {
      static_for<0, NS>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        const int Q = S.tq[s], P = S.tp[s];
        if (Q >= 0) {
          const int b = P >> 1, hp = P & 1;
          d4 acc = mf::zero4();
          const double* cf = sm + W::CF2 + b * NB;
          for (int k = b; k < NB; ++k) {
            const double coef = cf[k];
            const double* src = ex + uidx(Q, 2 * k + hp) * TSZ + G.g * TR + G.j;
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] += coef * src[4 * v * TR];
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) T[s][v] = G.ok[v] ? acc[v] : 0.0;
        }
      });
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
      // Sm = H C = H C0 + sigma2 H Q H' (d x d, plain index): the three blocks of its lower triangle, then its Cholesky,
      // W = L^-1, y = W z, z'Sm^-1 z and log det Sm (src/perform_step.jl:66)
      static_for<0, 3>([&](auto bc) {
        constexpr int blk = decltype(bc)::value;  // 0: (0,0)  1: (1,0)  2: (1,1)
        constexpr int ta = blk >= 1, tb = blk == 2;
        const int a_ = 16 * ta + G.j;                                   // A operand: row a of H (plain) = column pad_d(a) of Hs0
        const int acol = a_ < d ? pad_d(a_) : TR;                       // a zero column of Hs0 for the padding rows
        d4 acc = mf::zero4();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          acc = mf::mfma(hs0[(4 * ks + G.g) * LDP + acol], vp[(4 * ks + G.g) * LDP + G.j + 16 * tb], acc);
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
      factor_s(sm + W::SB11, sm + W::SB22, sm + W::SB21, sm + W::L21, wl, G);
      const int l = G.lane;
      double yv = 0.0, lg = 0.0;
      if (l < 32) {
        for (int b = 0; b <= l; ++b) yv += wl[l * LDP + b] * z[b];
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
    } else {
      // + sigma2 Q on the tiles (src/filtering.jl:35): Qt[Q/2][P/2] on the diagonal of the tiles with equal halves
      static_for<0, NS>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        const int Q = S.tq[s], P = S.tp[s];
        if (Q >= 0 && (Q & 1) == (P & 1)) {
          const double sq = sigma2_pred * pc.Qt[Q >> 1][P >> 1];
#pragma unroll
          for (int v = 0; v < 4; ++v) T[s][v] += (G.ok[v] && 4 * v + G.g == G.j) ? sq : 0.0;
        }
      });
    }
    __syncthreads();
    ODEF_MF_STAMP(10)
    G = fresh(G0);
    // per tile row R of the panels, one wavefront: C = C0 + sigma2 Q H', then V = C W' (in place) and K = V W
    for (int R = wave; R < NT; R += kMfWaves) {
      {
        const int bq = R >> 1, hr = R & 1;
        const double q0 = sigma2_pred * pc.Qt[bq][0], q1 = sigma2_pred * pc.Qt[bq][1] * h1;
        for (int e = G.lane; e < TR * d; e += 64) {  // (Q H')[r][a] = Qt[b][0] H0[a][i] + Qt[b][1] h1 [a == i],  r = (b, i)
          const int ii = e / d, a = e % d, i = TR * hr + ii;
          double x = q0 * hs0[pad_d(i) * LDP + pad_d(a)];
          if (a == i) x += q1;
          vp[(16 * R + ii) * LDP + a] += x;
        }
      }
      tv::lds_sync();
      d4 v0 = mf::zero4(), v1 = mf::zero4();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const double a = vp[(16 * R + G.j) * LDP + 4 * ks + G.g];
        v0 = mf::mfma(a, wl[G.j * LDP + 4 * ks + G.g], v0);
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
        k1 = mf::mfma(a, wl[(4 * ks + G.g) * LDP + 16 + G.j], k1);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        kp[(16 * R + 4 * v + G.g) * LDP + G.j] = k0[v];
        kp[(16 * R + 4 * v + G.g) * LDP + 16 + G.j] = k1[v];
      }
    }
    __syncthreads();
    ODEF_MF_STAMP(11)
    G = fresh(G0);
    // m = m^- - V y (src/filtering.jl:87), un-preconditioned (src/perform_step.jl:75);  T = S^- - V V'
    if (tid < D) {
      const int prow = 32 * (tid / d) + pad_d(tid % d);
      double s = mp[tid];
#pragma unroll 7
      for (int a = 0; a < d; ++a) s -= vp[prow * LDP + a] * y[a];
      m[tid] = tab[kTabPIJ + tid / d] * s;
    }
    if constexpr (!HELPER) rank_update(T, S, G0, vp, vp);  // bitwise symmetric on the diagonal tiles (same products, same order)
    __syncthreads();
    ODEF_MF_STAMP(12)
    if constexpr (!HELPER) tl_put(T, S, G0, tl, false);
    __syncthreads();
    ODEF_MF_STAMP(13)
    if constexpr (!HELPER) hproject(T, S, G0, hs0, tl, h1, vp);  // E = T H' over the V panel
    __syncthreads();
    ODEF_MF_STAMP(14)
    if constexpr (!HELPER) {
      rank_update(T, S, G0, vp, kp);  // S = T - E K'
      static_for<0, NS>([&](auto sc_) {  // un-precondition (src/perform_step.jl:73-75)
        constexpr int s = decltype(sc_)::value;
        if (S.tq[s] >= 0) T[s] *= tab[kTabPIPI + (S.tq[s] >> 1) * MAXNB + (S.tp[s] >> 1)];
      });
    }
    __syncthreads();
    ODEF_MF_STAMP(15)
  }

  template <bool HELPER>
  ODEF_MF_FN void save_record(const FilterParams& P, long i, long slot, double diffusion, const double* __restrict__ sm,
                              const d4 (&T)[NS], const Slots& S, const Geo& G, int tid) {
    const double* m = sm + W::MV;
    const size_t N = (size_t)P.N;
    if (tid < D) P.mean[((size_t)slot * D + tid) * N + i] = m[tid];
    if constexpr (!HELPER) {
      static_for<0, NS>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        const int Q = S.tq[s], Pc = S.tp[s];
        if (Q >= 0) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int a = Q * TR + 4 * v + G.g, b = Pc * TR + G.j;  // row <= column in the kept triangle
            if (G.ok[v] && a <= b) P.cov[((size_t)slot * TRI + tri(b, a)) * N + i] = T[s][v];
          }
        }
      });
    }
    if (tid == 0) P.diff[(size_t)slot * N + i] = diffusion;
  }

  // whole fixed-step solve of trajectory i
  template <bool HELPER>
  ODEF_MF_FN void run(const FilterParams& P, long i, int tid, double* __restrict__ sm) {
    double* m = sm + W::MV;
    double* sc = sm + W::SC;
    const size_t N = (size_t)P.N;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Geo G;
    G.lane = tid & 63;
    G.g = G.lane >> 4;
    G.j = tid & 15;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = 4 * v + G.g;
      G.ok[v] = (i < TR) && (G.j < TR);
      G.sym[v] = (i < G.j ? i : G.j) * TR + (i < G.j ? G.j : i);
    }
    Slots S;
    static_for<0, NS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      int qq = -1, pp = -1;
      static_for<0, kMfTileWaves>([&](auto wc) {
        constexpr int w = decltype(wc)::value;
        constexpr int cq = make_mf_own<NT>().Q[w][s], cp = make_mf_own<NT>().P[w][s];
        if (wave == w) {
          qq = cq;
          pp = cp;
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
    if (!HELPER && tid == 0) {  // Taylor-mode initial mean (src/state_initialization.jl), zero covariance
      double u0[d], m0[D];
      for (int a = 0; a < d; ++a) u0[a] = P.u0[(size_t)a * N + i];
      taylor_init<RHS, q>(u0, pl, m0);
      for (int k = 0; k < D; ++k) m[k] = m0[k];
      for (int k = 0; k < 8; ++k) sc[k] = 0.0;
      for (int a = 0; a < d; ++a) sm[W::UC + a] = u0[a];
    }
    for (int e = tid; e < 32; e += NTHR) sm[W::Z + e] = 0.0;  // the entries behind z[d-1] stay zero
    __syncthreads();
    if (P.everystep) save_record<HELPER>(P, i, 0, 0.0, sm, T, S, G, tid);
    for (long n = 0; n < P.nsteps; ++n) {
      const double* tab = P.ptab + (size_t)uniform_load(P.tab_idx + n) * kTabStride;
      step<HELPER>(P.pc, pl, tab, P.fixed_diffusion, (int)n, sm, T, S, G, tid, wave);
      if (P.everystep) save_record<HELPER>(P, i, n + 1, sc[4], sm, T, S, G, tid);
    }
    if (!P.everystep) save_record<HELPER>(P, i, 0, sc[4], sm, T, S, G, tid);
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
#undef ODEF_MF_FN
};

template <class RHS, int q, bool EK1>
__global__ __launch_bounds__(kMfBlock) void ek_filter_mfma_kernel(const FilterParams P) {
  using MF = MfmaFilter<RHS, q, EK1>;
  __shared__ double sm[MF::W::size];
  if (threadIdx.x >= 64 * kMfHelper)  // the helper wavefront: same barriers, its own code path and register allocation
    MF::template run<true>(P, (long)blockIdx.x, (int)threadIdx.x, sm);
  else
    MF::template run<false>(P, (long)blockIdx.x, (int)threadIdx.x, sm);
}

}  // namespace odef
#endif
