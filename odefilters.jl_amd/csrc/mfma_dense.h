// Dense FP64 algebra on the matrix cores for the workgroup-per-trajectory path (D = 28 (q+1) up to 168):
// everything is written as products of the form  C (op)= A' B  with BOTH operands "k-major" (row k of the operand matrix
// holds the k-th term of every output row / column), because that is what v_mfma_f64_16x16x4_f64 reads coalesced:
//
//     D(16x16) = A(16x4) B(4x16) + C      A: lane l holds A[i = l % 16][k = l / 16]
//                                         B: lane l holds B[k = l / 16][j = l % 16]
//                                         C/D: lane l holds D[i = 4 v + l / 16][j = l % 16], v = 0..3
//     (tools/mfma_layout_test.hip checks this on the hardware)
//
// so with P row-major and k-major, lane l of a fragment reads P[(k0 + l / 16) * ld + c0 + l % 16]: four runs of 128
// contiguous bytes per instruction, for the A side and the B side alike.  A 16x16 accumulator tile in the D layout can
// be fed back as the B (or A) operand of a K = 16 product without moving data: its register v IS the k-step v fragment.
//
//   B = U'U  (blocked Cholesky, upper factor, left-looking):   block row j of U from  B[j,:] - sum_{k<16j} U[k,j]' U[k,:]
//            then  U[j,:] <- W_j (.)  with  W_j = L_jj^-1  the explicitly inverted 16x16 diagonal block: the panel
//            "solve" is an MFMA product, there is no per-row substitution anywhere
//   U'U G' = Y'  two block sweeps of the same shape (the backward one reads L = U' k-major, kept beside U)
//   G M G'       two products with M symmetric
//
// EVERY function here is force-inlined.  Left to the inliner, the large ones (the gain phase, the Cholesky) became real
// device functions shared by kernels with different register budgets (the smoother at four workgroups per CU = 128
// registers, dense output and sampling at 256): a callee compiled for the wider budget then runs inside a wavefront that
// was allocated the narrower one -- a memory access fault at address 0 on the GPU, nothing at compile time.
//
// Matrices are padded to DP = 16 * ceil(D / 16) with zeros (the padding block of a matrix to be factorised gets a unit
// diagonal), so no fragment load or store is ever masked.
#pragma once
#include <hip/hip_runtime.h>
#include "team_vec.h"

namespace odef {
namespace mf {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int kB = 16;  // block size
#ifndef ODEF_MFMA_ATB_ROWS
#define ODEF_MFMA_ATB_ROWS 2
#endif
constexpr int kAtbRows = ODEF_MFMA_ATB_ROWS;  // block rows per accumulator tile set of the big products (x 3 block columns)

__device__ __attribute__((always_inline)) inline int lane64() { return (int)(threadIdx.x & 63u); }

// Tile and fragment accesses are BUFFER loads / stores: a matrix is a uniform base pointer (four scalar registers as a
// buffer resource), the lane's place in a tile one 32-bit offset register per leading dimension, the tile origin a scalar
// offset -- `buffer_load_dwordx2 v, v_off, s[rsrc], s_tile offen`.  Written as plain pointer arithmetic the compiler keeps
// a 64-bit address pair per access in vector registers (and spills them once a wavefront holds a tile column).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __attribute__((always_inline)) inline __amdgpu_buffer_rsrc_t mat_rsrc(const double* P) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)P, (short)0, 0x7fffffff, 0x00020000);
}
__device__ __attribute__((always_inline)) inline double ld8(const double* P, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(mat_rsrc(P), voff, soff, 0));
}
__device__ __attribute__((always_inline)) inline void st8(double* P, unsigned voff, unsigned soff, double x) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), mat_rsrc(P), voff, soff, 0);
}
__device__ __attribute__((always_inline)) inline unsigned lane_rc(int ld) {  // byte offset of (l / 16, l % 16)
  const unsigned l = threadIdx.x & 63u;
  return ((l >> 4) * (unsigned)ld + (l & 15u)) * 8u;
}
__device__ __attribute__((always_inline)) inline unsigned lane_cr(int ld) {  // byte offset of (l % 16, l / 16)
  const unsigned l = threadIdx.x & 63u;
  return ((l & 15u) * (unsigned)ld + (l >> 4)) * 8u;
}
__device__ __attribute__((always_inline)) inline unsigned tile_off(int ld, int r0, int c0) { return ((unsigned)r0 * (unsigned)ld + (unsigned)c0) * 8u; }
// fragment of a k-major row-major operand: rows k0 .. k0+3, columns c0 .. c0+15
__device__ __attribute__((always_inline)) inline double frag(const double* __restrict__ P, int ld, int k0, int c0) {
  return ld8(P, lane_rc(ld), tile_off(ld, k0, c0));
}
__device__ __attribute__((always_inline)) inline d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __attribute__((always_inline)) inline d4 zero4() { return d4{0.0, 0.0, 0.0, 0.0}; }

// tile (D layout) <-> row-major memory: element (4 v + l / 16, l % 16) at P[(r0 + 4 v + l / 16) * ld + c0 + l % 16]
__device__ __attribute__((always_inline)) inline d4 load_tile(const double* __restrict__ P, int ld, int r0, int c0) {
  const unsigned o = lane_rc(ld);
  d4 t;
#pragma unroll
  for (int v = 0; v < 4; ++v) t[v] = ld8(P, o, tile_off(ld, r0 + 4 * v, c0));
  return t;
}
__device__ __attribute__((always_inline)) inline void store_tile(double* __restrict__ P, int ld, int r0, int c0, d4 t) {
  const unsigned o = lane_rc(ld);
#pragma unroll
  for (int v = 0; v < 4; ++v) st8(P, o, tile_off(ld, r0 + 4 * v, c0), t[v]);
}
// the transposed tile: element (i, j) of t goes to P[(r0 + j) * ld + c0 + i]
__device__ __attribute__((always_inline)) inline void store_tile_t(double* __restrict__ P, int ld, int r0, int c0, d4 t) {
  const unsigned o = lane_cr(ld);
#pragma unroll
  for (int v = 0; v < 4; ++v) st8(P, o, tile_off(ld, r0, c0 + 4 * v), t[v]);
}

// acc[i][j] += sum_{k in [k0, k1)} A[k][m0 + 16 i + .] * B[k][n0 + 16 j + .]   (k0, k1 multiples of 4); one wavefront
template <int MB, int NB>
__device__ __attribute__((always_inline)) inline void wave_atb(const double* __restrict__ A, int lda, int m0,
                                                               const double* __restrict__ B, int ldb, int n0, int k0, int k1,
                                                               d4 (&acc)[MB][NB]) {
  // unrolled by two k-steps: the fragment loads of two steps (2 (MB + NB) loads) are in flight before the first product
  // needs them (the operands stream from L2 / HBM at ~1 us latency; load-wait-MFMA iterations are latency-bound)
#pragma unroll 2
  for (int k = k0; k < k1; k += 4) {
    double a[MB], b[NB];
#pragma unroll
    for (int i = 0; i < MB; ++i) a[i] = frag(A, lda, k, m0 + kB * i);
#pragma unroll
    for (int j = 0; j < NB; ++j) b[j] = frag(B, ldb, k, n0 + kB * j);
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = mfma(a[i], b[j], acc[i][j]);
  }
}
// the same for one block row (m0) against NT tiles at arbitrary column offsets n0[t] (one A fragment feeds NT products)
template <int NT>
__device__ __attribute__((always_inline)) inline void wave_atb_cols(const double* __restrict__ A, int lda, int m0,
                                                                    const double* __restrict__ B, int ldb, const int (&n0)[NT],
                                                                    int k0, int k1, d4 (&acc)[NT]) {
#pragma unroll 8
  for (int k = k0; k < k1; k += 4) {
    const double a = frag(A, lda, k, m0);
    double b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = frag(B, ldb, k, n0[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma(a, b[t], acc[t]);
  }
}

// two block rows (m0, m1) against the same NT column tiles: every B fragment feeds two products
template <int NT>
__device__ __attribute__((always_inline)) inline void wave_atb_cols2(const double* __restrict__ A, int lda, int m0, int m1,
                                                                     const double* __restrict__ B, int ldb, const int (&n0)[NT],
                                                                     int k0, int k1, d4 (&acc0)[NT], d4 (&acc1)[NT]) {
#pragma unroll 4
  for (int k = k0; k < k1; k += 4) {
    const double a0 = frag(A, lda, k, m0), a1 = frag(A, lda, k, m1);
    double b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = frag(B, ldb, k, n0[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc0[t] = mfma(a0, b[t], acc0[t]);
      acc1[t] = mfma(a1, b[t], acc1[t]);
    }
  }
}

// One MB x NB block tile of  C = [Cin -] A'B  (K = kdim rows of A and B), computed and stored by one wavefront.
template <int MB, int NB, bool SUB>
__device__ __attribute__((always_inline)) inline void wave_atb_store(const double* __restrict__ A, int lda, int m0, const double* __restrict__ B, int ldb, int n0,
                                      int kdim, const double* __restrict__ Cin, double* __restrict__ C, int ldc) {
  d4 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = zero4();
  wave_atb<MB, NB>(A, lda, m0, B, ldb, n0, 0, kdim, acc);
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      d4 t = acc[i][j];
      if constexpr (SUB) t = load_tile(Cin, ldc, m0 + kB * i, n0 + kB * j) - t;
      store_tile(C, ldc, m0 + kB * i, n0 + kB * j, t);
    }
}
template <int NB, bool SUB>
__device__ __attribute__((always_inline)) inline void wave_atb_rows(int mb, const double* __restrict__ A, int lda, int m0, const double* __restrict__ B, int ldb,
                                     int n0, int kdim, const double* __restrict__ Cin, double* __restrict__ C, int ldc) {
  if (mb == 3) wave_atb_store<3, NB, SUB>(A, lda, m0, B, ldb, n0, kdim, Cin, C, ldc);
  else if (mb == 2) wave_atb_store<2, NB, SUB>(A, lda, m0, B, ldb, n0, kdim, Cin, C, ldc);
  else wave_atb_store<1, NB, SUB>(A, lda, m0, B, ldb, n0, kdim, Cin, C, ldc);
}
// C = [Cin -] A'B over block rows [ib0, ib1) x block columns [jb0, jb1) by the wavefronts of a workgroup: strips of up
// to 3 block columns go round-robin over the wavefronts, each strip in chunks of up to 3 block rows (9 accumulator
// tiles per wavefront; per k-step 6 fragment loads feed 9 MFMAs; small enough for two workgroups per CU).  No two wavefronts touch the same block, so Cin may
// alias C.
template <bool SUB>
__device__ __attribute__((always_inline)) inline void wg_atb(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, int kdim,
                              const double* __restrict__ Cin, double* __restrict__ C, int ldc, int ib0, int ib1, int jb0, int jb1) {
  const int wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6);
  int strip = 0;
  for (int jb = jb0; jb < jb1; jb += 3, ++strip) {
    if (strip % nwaves != wave) continue;
    const int nb = (jb1 - jb) < 3 ? (jb1 - jb) : 3;
    for (int ib = ib0; ib < ib1; ib += kAtbRows) {
      const int mb = (ib1 - ib) < kAtbRows ? (ib1 - ib) : kAtbRows;
      if (nb == 3) wave_atb_rows<3, SUB>(mb, A, lda, ib * kB, B, ldb, jb * kB, kdim, Cin, C, ldc);
      else if (nb == 2) wave_atb_rows<2, SUB>(mb, A, lda, ib * kB, B, ldb, jb * kB, kdim, Cin, C, ldc);
      else wave_atb_rows<1, SUB>(mb, A, lda, ib * kB, B, ldb, jb * kB, kdim, Cin, C, ldc);
    }
  }
}

// ---- the 16 x 16 diagonal block: Cholesky D = L L' and W = L^-1, by one wavefront -------------------------------------
// Input: the symmetric block in LDS `blk` [16][16] (lower triangle referenced).  Output: `lw` [16][16] = L (lower,
// row-major; skipped when null), `w` [16][16] = W = L^-1 (lower).  Row-lane Cholesky on the first 16 lanes with DPP broadcasts (the other
// 48 lanes of the wavefront repeat it on the same data), then the rows of W = L^-1 by back substitution, again with DPP.
// A non-positive pivot zeroes its column (semi-definite rule of ek_math.h); its reciprocal is taken as 0.
__device__ __attribute__((always_inline)) inline void diag_block_factor(double* __restrict__ blk, double* __restrict__ lw, double* __restrict__ w, int ldw = kB) {
  const int r = tv::lane();
  double row[kB];
#pragma unroll
  for (int c = 0; c < kB; ++c) row[c] = blk[r * kB + c];
  double dinv_r = 0.0;  // 1 / L[r][r] of the own row
  static_for<0, kB>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double piv = tv::bcast<k>(row[k]);
    const bool ok = piv > 0.0;
    double root, rroot;
    sqrt_and_rsqrt(ok ? piv : 1.0, root, rroot);
    root = ok ? root : 0.0;
    rroot = ok ? rroot : 0.0;
    const double colk = row[k];          // lane j: D[j][k] of the current Schur complement
    const double lik = colk * rroot;     // L[r][k] for r >= k
    if constexpr (k + 1 < kB) tv::fb_cols<true, k + 1, kB - k - 1>(row, lik, lik);  // row[j] -= L[r][k] L[j][k], j > k
    row[k] = lik;
    dinv_r = (r == k) ? rroot : dinv_r;
    (void)root;
  });
  if (lw) {
#pragma unroll
    for (int c = 0; c < kB; ++c) lw[r * kB + c] = (c <= r) ? row[c] : 0.0;
  }
  blk[kB * kB + r] = dinv_r;  // reciprocals of the diagonal, behind the block
  // W = L^-1, row r in lane r:  W L = I  =>  W[r][k] = (delta_rk - sum_{k' > k} W[r][k'] L[k'][k]) / L[k][k], k descending.
  // L[k'][k] is lane k''s register row[k]: one fused DPP broadcast-FMA per term, 120 in all, no LDS round trip (the
  // column-per-lane form it replaces read L back from LDS one dependent ds_read at a time: 8 000 of the 12 000 cycles).
  double acc[kB];
#pragma unroll
  for (int c = 0; c < kB; ++c) acc[c] = (c == r) ? 1.0 : 0.0;
  static_for<0, kB>([&](auto kc) {
    constexpr int k = kB - 1 - decltype(kc)::value;
    const double wk = acc[k] * tv::bcast<k>(dinv_r);
    acc[k] = wk;
    if constexpr (k > 0) tv::fb_rows<true, k, k>(acc, row, wk);
  });
#pragma unroll
  for (int c = 0; c < kB; ++c) w[r * ldw + c] = acc[c];
  tv::lds_sync();
}

// The same factorisation where only ONE solve is wanted: returns y = L^-1 t (lane r: y[r], t[r]) instead of W = L^-1 -- forward
// substitution by columns in the row-lane layout (lane k holds row k of L): y_j = t_j / L[j][j] in lane j, then every lane k > j
// takes t_k -= L[k][j] y_j with one fused DPP broadcast-FMA.  48 instructions where the inverse costs ~450 (16 broadcasts, 120
// broadcast-FMAs, their wait states, 16 stores) plus the product with it.  A non-positive pivot gives y_j = 0 (its column of L
// is zero: the semi-definite rule above).
__device__ __attribute__((always_inline)) inline double diag_block_factor_solve(const double* __restrict__ blk, double t) {
  const int r = tv::lane();
  double row[kB];
#pragma unroll
  for (int c = 0; c < kB; ++c) row[c] = blk[r * kB + c];
  double dinv_r = 0.0;
  static_for<0, kB>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double piv = tv::bcast<k>(row[k]);
    const bool ok = piv > 0.0;
    double root, rroot;
    sqrt_and_rsqrt(ok ? piv : 1.0, root, rroot);
    rroot = ok ? rroot : 0.0;
    const double lik = row[k] * rroot;
    if constexpr (k + 1 < kB) tv::fb_cols<true, k + 1, kB - k - 1>(row, lik, lik);
    row[k] = lik;
    dinv_r = (r == k) ? rroot : dinv_r;
    (void)root;
  });
  static_for<0, kB>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const double yj = t * dinv_r;  // (final in lane j; the lanes below j are done, those above still change)
    if constexpr (j + 1 < kB) {
      double tn = t;
      tv::fnma_bc<j>(tn, yj, row[j]);  // t_k -= y_j L[k][j]
      t = (r > j) ? tn : t;
    }
  });
  return t * dinv_r;
}

// a-fragment of the K = 16 product  W (.)  /  W' (.)  from the 16 x 16 block W in LDS (row-major):
//   TRANS == false:  out = W  R  ->  A'[n][k'] = W[k'][n]        TRANS == true:  out = W' R  ->  A'[n][k'] = W[n][k']
template <bool TRANS>
__device__ __attribute__((always_inline)) inline double wfrag(const double* __restrict__ w, int kk) {
  const int l = lane64();
  return TRANS ? w[(4 * kk + (l >> 4)) * kB + (l & 15)] : w[(l & 15) * kB + 4 * kk + (l >> 4)];
}
// out = W r  (or W' r) for a 16 x 16 tile r in the D layout: its register v is the B fragment of k-step v
template <bool TRANS>
__device__ __attribute__((always_inline)) inline d4 apply_w(const double* __restrict__ w, d4 r) {
  d4 o = zero4();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) o = mfma(wfrag<TRANS>(w, kk), r[kk], o);
  return o;
}

// LDS layout of the factorisation scratch (doubles)
template <int DPB>
struct CholLds {
  static constexpr int blk = 0;                    // 16 x 16 block + 16 reciprocals
  static constexpr int lw = blk + kB * kB + kB;    // its factor L
  static constexpr int w = lw + kB * kB;           // W_j = L_jj^-1, j = 0..DPB-1 (kept for the two sweeps)
  static constexpr int size = w + DPB * kB * kB;
};

// Blocked Cholesky  B = U'U  IN PLACE in the upper triangle of Bm (full symmetric storage on entry; the lower blocks
// are not touched), left-looking by block rows; Lm receives L = U' (lower, row-major) for the backward sweep.
// A workgroup of 4 wavefronts; `lds`: CholLds<DPB>::size doubles.
template <int DPB>
__device__ __attribute__((always_inline)) inline void wg_cholesky_upper(double* __restrict__ Bm, double* __restrict__ Lm, int ld, double* __restrict__ lds) {
  using LL = CholLds<DPB>;
  const int wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6);
  constexpr int MAXT = (DPB + 3) / 4;  // tiles of a block row per wavefront (4 wavefronts)
  for (int j = 0; j < DPB; ++j) {
    d4 t[MAXT];
    // 1. tiles (j, mb), mb = j + wave, j + wave + nwaves, ...:  B[j, mb] - sum_{k < 16 j} U[k, j]' U[k, mb]
    //    (a tile index past the last block is clamped: computed on valid memory, never stored)
    int n0[MAXT];
#pragma unroll
    for (int s = 0; s < MAXT; ++s) {
      const int mb = j + wave + s * nwaves;
      n0[s] = (mb < DPB ? mb : DPB - 1) * kB;
      t[s] = zero4();
    }
    wave_atb_cols<MAXT>(Bm, ld, j * kB, Bm, ld, n0, 0, j * kB, t);
#pragma unroll
    for (int s = 0; s < MAXT; ++s) {
      const int mb = j + wave + s * nwaves;
      if (mb < DPB) {
        t[s] = load_tile(Bm, ld, j * kB, mb * kB) - t[s];
        if (mb == j) {  // the diagonal block goes to LDS for the factorisation
          const int l = lane64();
#pragma unroll
          for (int v = 0; v < 4; ++v) lds[LL::blk + (4 * v + (l >> 4)) * kB + (l & 15)] = t[s][v];
        }
      }
    }
    __syncthreads();
    if (wave == 0) diag_block_factor(lds + LL::blk, lds + LL::lw, lds + LL::w + j * kB * kB);
    __syncthreads();
    // 2. U[j, mb] = W_j (tile);  the diagonal tile is L_jj' (its strictly lower part is rounding noise: zeroed)
    const double* wj = lds + LL::w + j * kB * kB;
#pragma unroll
    for (int s = 0; s < MAXT; ++s) {
      const int mb = j + wave + s * nwaves;
      if (mb < DPB) {
        d4 u = apply_w<false>(wj, t[s]);
        if (mb == j) {
          const int l = lane64();
#pragma unroll
          for (int v = 0; v < 4; ++v) u[v] = (4 * v + (l >> 4) > (l & 15)) ? 0.0 : u[v];
        }
        store_tile(Bm, ld, j * kB, mb * kB, u);
        store_tile_t(Lm, ld, mb * kB, j * kB, u);
      }
    }
    __syncthreads();  // block row j of U is in memory for the rows below
  }
}

// Gt <- (U'U)^-1 Yt in place (DP x DP right-hand sides, row-major; U upper in Um, L = U' in Lm, the inverted diagonal
// blocks in LDS from wg_cholesky_upper): forward sweep U' Z = Yt by block rows top-down, backward sweep U Gt = Z bottom-up.
template <int DPB>
__device__ __attribute__((always_inline)) inline void wg_solve_upper(const double* __restrict__ Um, const double* __restrict__ Lm, double* __restrict__ Yt, int ld,
                                      const double* __restrict__ lds) {
  using LL = CholLds<DPB>;
  const int wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6);
  constexpr int MAXT = (DPB + 3) / 4;
  int n0[MAXT];
#pragma unroll
  for (int s = 0; s < MAXT; ++s) {
    const int cb = wave + s * nwaves;
    n0[s] = (cb < DPB ? cb : DPB - 1) * kB;
  }
  // Two block rows per step: the rows above (below) are streamed ONCE for both -- the re-reads of the finished rows are
  // the traffic of a left-looking sweep (every tile of Z once per later block row: 1.2 MB per sweep at DPB = 11) -- and the
  // second row picks up the first one's result from the accumulator registers (a D-layout tile is a K = 16 B operand).
  for (int j = 0; j < DPB; j += 2) {  // Z_j = W_j (Yt_j - sum_{k < 16 j} U[k, j]' Z[k, :])
    const bool two = j + 1 < DPB;
    const int j1 = two ? j + 1 : j;
    d4 acc0[MAXT], acc1[MAXT];
#pragma unroll
    for (int s = 0; s < MAXT; ++s) acc0[s] = acc1[s] = zero4();
    wave_atb_cols2<MAXT>(Um, ld, j * kB, j1 * kB, Yt, ld, n0, 0, j * kB, acc0, acc1);
    const double* w0 = lds + LL::w + j * kB * kB;
    const double* w1 = lds + LL::w + j1 * kB * kB;
#pragma unroll
    for (int s = 0; s < MAXT; ++s) {
      const int cb = wave + s * nwaves;
      if (cb < DPB) {
        const d4 z0 = apply_w<false>(w0, load_tile(Yt, ld, j * kB, cb * kB) - acc0[s]);
        store_tile(Yt, ld, j * kB, cb * kB, z0);
        if (two) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc1[s] = mfma(frag(Um, ld, j * kB + 4 * ks, j1 * kB), z0[ks], acc1[s]);
          store_tile(Yt, ld, j1 * kB, cb * kB, apply_w<false>(w1, load_tile(Yt, ld, j1 * kB, cb * kB) - acc1[s]));
        }
      }
    }
    __syncthreads();
  }
  for (int j = DPB - 1; j >= 0; j -= 2) {  // Gt_j = W_j' (Z_j - sum_{k >= 16 (j+1)} L[k, j]' Gt[k, :])
    const bool two = j >= 1;
    const int j1 = two ? j - 1 : j;
    d4 acc0[MAXT], acc1[MAXT];
#pragma unroll
    for (int s = 0; s < MAXT; ++s) acc0[s] = acc1[s] = zero4();
    wave_atb_cols2<MAXT>(Lm, ld, j * kB, j1 * kB, Yt, ld, n0, (j + 1) * kB, DPB * kB, acc0, acc1);
    const double* w0 = lds + LL::w + j * kB * kB;
    const double* w1 = lds + LL::w + j1 * kB * kB;
#pragma unroll
    for (int s = 0; s < MAXT; ++s) {
      const int cb = wave + s * nwaves;
      if (cb < DPB) {
        const d4 g0 = apply_w<true>(w0, load_tile(Yt, ld, j * kB, cb * kB) - acc0[s]);
        store_tile(Yt, ld, j * kB, cb * kB, g0);
        if (two) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc1[s] = mfma(frag(Lm, ld, j * kB + 4 * ks, j1 * kB), g0[ks], acc1[s]);
          store_tile(Yt, ld, j1 * kB, cb * kB, apply_w<true>(w1, load_tile(Yt, ld, j1 * kB, cb * kB) - acc1[s]));
        }
      }
    }
    __syncthreads();
  }
}

// ---- register-resident variants (round 2, second half): the right-hand sides / the B operand stay in the accumulator
// registers of the wavefront that owns their tile COLUMN, for the whole sweep / product.  Columns are independent in
// both, so there is no barrier inside; the shared operand (U, L, M, Z) streams from L2 as k-major fragments.  With four
// wavefronts and 256 registers a wavefront can hold two tile columns (2 x DPB tiles): DPB = 11 columns go in two passes
// (columns 0..7 two per wavefront, then 8..10 one per wavefront).
// MEASURED (tools/mfma_dense_test.hip and the D = 168 smoother, -DODEF_SMOOTH_RR): correct to 2e-15 and SLOWER than the
// left-looking forms above -- 509 against 410 us for one workgroup, 529 against 352 ms for 2 048 trajectories x 64 steps.
// The barriers it removes were not the cost: with two columns resident a wavefront has 22 dependent accumulator chains
// but no registers left to keep the next k-major fragments in flight, so every 4-load group is waited for (L2 latency per
// block pair), and the operand traffic is the same (every wavefront streams all of U / L / M per pass).  Kept for the
// record and for a layout with fewer, wider wavefronts; not used by default.  (One column per wavefront and pass, small
// enough for four workgroups per CU, was slower still: 483 against 356 us per sweep pair -- every wavefront then streams
// all of U and L once per COLUMN, 2.7 MB per step, where the left-looking form shares each tile between the four
// wavefronts of the block row through the L1.)
template <int DPB>
struct ColPass {
  static constexpr int kPasses = (DPB + 7) / 8;
  // columns of (wave, pass): c0 always valid if n >= 1
  __device__ static inline int ncols(int wave, int pass) {
    const int first = 8 * pass, rest = DPB - first;
    if (rest >= 8) return 2;
    // the last pass: `rest` (< 8) columns, one per wavefront first, then a second one
    return (wave < rest ? 1 : 0) + (wave + 4 < rest ? 1 : 0);
  }
  __device__ static inline int col(int wave, int pass, int t) {
    const int first = 8 * pass, rest = DPB - first;
    if (rest >= 8) return first + 2 * wave + t;
    return first + wave + 4 * t;
  }
};

// Gt <- (U'U)^-1 Yt in place, as wg_solve_upper, but RIGHT-looking with the right-hand sides resident: per pass a
// wavefront loads its <= 2 tile columns of Yt (2 x DPB tiles), runs the forward sweep (Z_j = W_j acc_j, then
// acc_j' -= U[j, j']' Z_j for j' > j), the backward sweep (Gt_j = W_j' acc_j, then acc_j' -= U[j', j] Gt_j for j' < j,
// read k-major from L = U'), and stores them.  No barrier: 22 block steps of the left-looking form each ended in one.
template <int DPB>
__device__ __attribute__((always_inline)) inline void wg_solve_upper_rr(const double* __restrict__ Um, const double* __restrict__ Lm, double* __restrict__ Yt, int ld,
                                         const double* __restrict__ lds) {
  using LL = CholLds<DPB>;
  const int wave = (int)(threadIdx.x >> 6);
  for (int pass = 0; pass < ColPass<DPB>::kPasses; ++pass) {
    const int nc = ColPass<DPB>::ncols(wave, pass);
    if (nc == 0) continue;
    const int c0 = ColPass<DPB>::col(wave, pass, 0) * kB, c1 = (nc > 1 ? ColPass<DPB>::col(wave, pass, 1) : ColPass<DPB>::col(wave, pass, 0)) * kB;
    d4 acc[DPB][2];
#pragma unroll
    for (int j = 0; j < DPB; ++j) {
      acc[j][0] = load_tile(Yt, ld, j * kB, c0);
      acc[j][1] = load_tile(Yt, ld, j * kB, c1);  // (a lone column is carried twice: same code, no branches in the sweeps)
    }
    static_for<0, DPB>([&](auto jc) {  // forward
      constexpr int j = decltype(jc)::value;
      const double* wj = lds + LL::w + j * kB * kB;
      const d4 z0 = apply_w<false>(wj, acc[j][0]), z1 = apply_w<false>(wj, acc[j][1]);
      acc[j][0] = z0;
      acc[j][1] = z1;
      static_for<j + 1, DPB>([&](auto jpc) {
        constexpr int jp = decltype(jpc)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const double a = -frag(Um, ld, j * kB + 4 * ks, jp * kB);
          acc[jp][0] = mfma(a, z0[ks], acc[jp][0]);
          acc[jp][1] = mfma(a, z1[ks], acc[jp][1]);
        }
      });
    });
    static_for<0, DPB>([&](auto jc) {  // backward
      constexpr int j = DPB - 1 - decltype(jc)::value;
      const double* wj = lds + LL::w + j * kB * kB;
      const d4 g0 = apply_w<true>(wj, acc[j][0]), g1 = apply_w<true>(wj, acc[j][1]);
      acc[j][0] = g0;
      acc[j][1] = g1;
      static_for<0, j>([&](auto jpc) {
        constexpr int jp = decltype(jpc)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const double a = -frag(Lm, ld, j * kB + 4 * ks, jp * kB);
          acc[jp][0] = mfma(a, g0[ks], acc[jp][0]);
          acc[jp][1] = mfma(a, g1[ks], acc[jp][1]);
        }
      });
    });
#pragma unroll
    for (int j = 0; j < DPB; ++j) {
      store_tile(Yt, ld, j * kB, c0, acc[j][0]);
      if (nc > 1) store_tile(Yt, ld, j * kB, c1, acc[j][1]);
    }
  }
}

// C = A'B (DPB x DPB tiles, all row-major with the same leading dimension) with the tile columns of B resident in the
// registers of their wavefront: per pass a wavefront loads its <= 2 tile columns of B once and produces the same columns
// of C, two tile rows at a time, from k-major fragments of A.
template <int DPB>
__device__ __attribute__((always_inline)) inline void wg_atb_rescols(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int ld) {
  const int wave = (int)(threadIdx.x >> 6);
  for (int pass = 0; pass < ColPass<DPB>::kPasses; ++pass) {
    const int nc = ColPass<DPB>::ncols(wave, pass);
    if (nc == 0) continue;
    const int c0 = ColPass<DPB>::col(wave, pass, 0) * kB, c1 = (nc > 1 ? ColPass<DPB>::col(wave, pass, 1) : ColPass<DPB>::col(wave, pass, 0)) * kB;
    d4 b[DPB][2];
#pragma unroll
    for (int k = 0; k < DPB; ++k) {
      b[k][0] = load_tile(B, ld, k * kB, c0);
      b[k][1] = load_tile(B, ld, k * kB, c1);
    }
    for (int r = 0; r < DPB; r += 2) {
      const int r1 = r + 1 < DPB ? r + 1 : r;
      d4 o00 = zero4(), o01 = zero4(), o10 = zero4(), o11 = zero4();
      static_for<0, DPB>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const double a0 = frag(A, ld, k * kB + 4 * ks, r * kB), a1 = frag(A, ld, k * kB + 4 * ks, r1 * kB);
          o00 = mfma(a0, b[k][0][ks], o00);
          o01 = mfma(a0, b[k][1][ks], o01);
          o10 = mfma(a1, b[k][0][ks], o10);
          o11 = mfma(a1, b[k][1][ks], o11);
        }
      });
      store_tile(C, ld, r * kB, c0, o00);
      if (nc > 1) store_tile(C, ld, r * kB, c1, o01);
      if (r1 != r) {
        store_tile(C, ld, r1 * kB, c0, o10);
        if (nc > 1) store_tile(C, ld, r1 * kB, c1, o11);
      }
    }
  }
}

}  // namespace mf
}  // namespace odef
