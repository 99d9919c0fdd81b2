// The covariance products of one RTS step on chip (src/smoothing.jl:45-63 in the textbook form, see smooth_mfma.h):
//   R = G M G' = Z' G',  Z = M G'        (M = P Sigma^s_+ P - B symmetric, G' = B^-1 A X)
// for the workgroup that has just finished the two sweeps (ek_kernels.h, rts_smooth_sweeps_kernel): wavefront c holds tile
// column c of G' in its accumulators (DPB tiles of 16 x 16, register v of a tile = rows 4 v + l / 16, column l % 16 -- which
// IS the B operand of a K = 16 product).  Nothing of G' or Z goes through memory:
//
//   M        upper tiles in the LDS the factor has left (swizzled, unpadded: 2 KB per tile, both the direct and the
//            transposed fragment read are bank-conflict free -- at DPB = 11 there is no room for padded rows beside a row of Z)
//   row i    Z[i, c] = sum_k M[i, k] G'[k, c]          DPB tile products per wavefront; the tile goes to the LDS row buffer
//            R[c', c] += Z[i, c']' G'[i, c]            for the (DPB + 1) / 2 tiles c' = c, c + 1, ... (mod DPB) this wavefront
//                                                      owns: every unordered pair of tile columns has exactly one owner
//
// 3 DPB^3 / 2 tile products instead of the 2 DPB^3 of two full products, no Z and no G' in the workspace (the workspace
// kernel's products read 22 operand tiles from L2 / HBM per result tile; PMC, profiles/r03_pleiades_smoother_pmc.json:
// 8.0 MB per trajectory and record).
#pragma once
#include "mfma_dense.h"

namespace odef {
namespace oc {

using mf::d4;

template <int DPB>
struct Products {
  static constexpr int NTU = DPB * (DPB + 1) / 2;
  static constexpr int WMAX = DPB / 2 + 1;                     // tiles of R per wavefront (the last one only where the pair has no other owner)
  static constexpr int kM = 0, kMsize = NTU * 256;             // M, upper tiles, swizzled
  static constexpr int kZ = kMsize;                            // row buffer(s): DPB tiles of 256 doubles, row-major
  static constexpr int kLdsDoubles = 160 * 1024 / 8;
  static constexpr int kVecRoom = 4 * DPB * 16;                // what the caller keeps behind (vectors)
  static constexpr int NBUF = (kMsize + 2 * DPB * 256 + kVecRoom <= kLdsDoubles) ? 2 : 1;
  static constexpr int size = kMsize + NBUF * DPB * 256;
  __host__ __device__ static constexpr int tix(int j, int jp) { return j * DPB - j * (j - 1) / 2 + (jp - j); }  // upper tile (j, jp >= j)
  // number of tiles of R wavefront c owns: c' = c + w (mod DPB), w < owned(c)
  __host__ __device__ static constexpr int owned(int c) { return (DPB % 2 == 1 || c < DPB / 2) ? DPB / 2 + 1 : DPB / 2; }
  // element (r, c) of a swizzled tile
  __host__ __device__ static constexpr int sw(int r, int c) { return r * 16 + (c ^ ((r >> 1) << 1)); }
};

// one 16 x 16 tile of a tile-major matrix (256 contiguous doubles at `tile`) in the accumulator layout: register v = rows
// 4 v + l / 16, column l % 16 -- 512 contiguous bytes per load instruction
__device__ __attribute__((always_inline)) inline d4 load_tile_major(const double* __restrict__ tile) {
  const int l = (int)threadIdx.x & 63;
  d4 t;
#pragma unroll
  for (int v = 0; v < 4; ++v) t[v] = tile[64 * v + l];
  return t;
}

__device__ __attribute__((always_inline)) inline void store_tile_major(double* __restrict__ tile, const d4& t) {
  const int l = (int)threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < 4; ++v) tile[64 * v + l] = t[v];
}

// M (symmetric, tile-major with the tiles of a tile column one after the other: tile (tr, tc) at (tc DPB + tr) 256; upper tiles
// present) -> LDS, upper tiles in row order; the workgroup has DPB wavefronts, wavefront w takes the tiles w, w + DPB, ...: all
// its loads in flight, then the stores
template <int DPB>
__device__ __attribute__((always_inline)) inline void load_m(const double* __restrict__ MM, double* __restrict__ lds) {
  using Pr = Products<DPB>;
  constexpr int PER = (Pr::NTU + DPB - 1) / DPB;
  const int tid = (int)threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  d4 x[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int t = wave + u * DPB;
    int j = 0, rest = t;
    while (rest >= DPB - j) {  // block row j from t by counting down the row lengths (wavefront-uniform)
      rest -= DPB - j;
      ++j;
    }
    if (t < Pr::NTU) x[u] = load_tile_major(MM + ((j + rest) * DPB + j) * 256);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int t = wave + u * DPB;
    if (t < Pr::NTU) {
      double* dst = lds + Pr::kM + t * 256;
#pragma unroll
      for (int v = 0; v < 4; ++v) dst[Pr::sw(4 * v + (l >> 4), l & 15)] = x[u][v];
    }
  }
}

// acc[k] = G'[k, c] (c = this wavefront's tile column), M in LDS (load_m, synchronised by the caller).  On return r[w] = R[c + w mod DPB, c]
// for w < owned(c).  Contains workgroup barriers: every wavefront of the workgroup must call it.
// KL: k-steps (of 4 rows) of the LAST tile row that hold state components -- the rows of G' behind the state dimension are
// zero, and the products skip them (D = 168: 2 of 4)
template <int DPB, int KL = 4>
__device__ __attribute__((always_inline)) inline void gmgt(const d4 (&acc)[DPB], double* __restrict__ lds, d4 (&r)[Products<DPB>::WMAX]) {
  using Pr = Products<DPB>;
  const int tid = (int)threadIdx.x, c = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const int n_own = Pr::owned(c);
  // lane offsets of the two fragment reads of a swizzled tile: direct A[m = l % 16][k = 4 kk + l / 16], transposed A'[..]
  int off_d[4], off_t[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    off_d[kk] = Pr::sw(l & 15, 4 * kk + (l >> 4));
    off_t[kk] = Pr::sw(4 * kk + (l >> 4), l & 15);
    // (laundered: inside a caller's loop the compiler would otherwise hoist one address register per tile and k-step out
    // of it and spill them by the hundred)
    asm volatile("" : "+v"(off_d[kk]), "+v"(off_t[kk]));
  }
  int off_z = (l >> 4) * 16 + (l & 15);  // row buffer: element (4 kk + l / 16, l % 16) = off_z + 64 kk
  asm volatile("" : "+v"(off_z));
#pragma unroll
  for (int w = 0; w < Pr::WMAX; ++w) r[w] = mf::zero4();
  static_for<0, DPB>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    d4 z = mf::zero4();
    static_for<0, DPB>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int t = i <= k ? Pr::tix(i, k) : Pr::tix(k, i);
      const double* m = lds + Pr::kM + t * 256;
#pragma unroll
      for (int kk = 0; kk < (k == DPB - 1 ? KL : 4); ++kk) z = mf::mfma(m[i <= k ? off_d[kk] : off_t[kk]], acc[k][kk], z);
      asm volatile("" ::: "memory");  // (keeps the compiler from hoisting, and spilling, the fragment reads of all later tiles)
    });
    int oz = off_z;  // (a fresh copy per row: the addresses of the owned tiles live for one row, not for the whole product)
    asm volatile("" : "+v"(oz));
    double* zrow = lds + Pr::kZ + (Pr::NBUF == 2 ? (i & 1) * DPB * 256 : 0) + oz;
    if constexpr (Pr::NBUF == 1 && i > 0) __syncthreads();  // the readers of row i - 1 are done with the buffer
#pragma unroll
    for (int v = 0; v < 4; ++v) zrow[c * 256 + 64 * v] = z[v];
    __syncthreads();
#pragma unroll
    for (int w = 0; w < Pr::WMAX; ++w) {
      if (w < n_own) {
        const int cw = c + w < DPB ? c + w : c + w - DPB;
        const double* zt = zrow + cw * 256;
#pragma unroll
        for (int kk = 0; kk < (i == DPB - 1 ? KL : 4); ++kk) r[w] = mf::mfma(zt[64 * kk], acc[i][kk], r[w]);
        asm volatile("" ::: "memory");
      }
    }
  });
}

// t[col] = sum_row G'[row, col] x[row] for the 16 columns of this wavefront's tile column (x in LDS, DPB * 16 entries, zero
// behind the state dimension); valid in every lane (lane l: column l % 16).  Summation order: rows r = g mod 4 in
// increasing order per class g, then (s0 + s1) + (s2 + s3) -- mfma_gain_phase (smooth_mfma.h) sums the same way.
template <int DPB>
__device__ __attribute__((always_inline)) inline double gt_times(const d4 (&acc)[DPB], const double* __restrict__ x) {
  const int l = (int)threadIdx.x & 63;
  double t = 0.0;
#pragma unroll
  for (int j = 0; j < DPB; ++j)
#pragma unroll
    for (int v = 0; v < 4; ++v) t = fma(acc[j][v], x[16 * j + 4 * v + (l >> 4)], t);
  t += __shfl_xor(t, 16);
  t += __shfl_xor(t, 32);
  return t;
}

}  // namespace oc
}  // namespace odef
