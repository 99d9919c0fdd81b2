// "One value per lane of a 16-lane team" vector type for the row-per-lane kernels (rows_filter.h, rows_smooth.h):
// 16 lanes -- one DPP row of a gfx950 wavefront -- own ONE trajectory, lane r keeps row r of every D x D matrix
// (D <= 16) and component r of every D-vector in registers; 4 trajectories per wavefront.
//
// The algorithms are written once in terms of
//   TV      a value that differs from lane to lane (row entries, vector components)
//   double  a TEAM-UNIFORM value (the d x d measurement algebra, step sizes, the controller) -- on the device every
//           lane of the team carries its own identical copy
// and the cross-lane primitives below:
//   device (gfx950):  TV is a plain double in a VGPR pair.  bcast<K> is ONE v_mov_b64_dpp row_newbcast:K, and
//                     acc += bcast<K>(src) * b is ONE v_fmac_f64_dpp row_newbcast:K -- the only DPP control the FP64
//                     ALU accepts, and exactly the one a row-per-lane factorisation needs (pivot row / pivot column to
//                     everybody).  Measured (tools/dpp_bench.hip, profiles/r02_dpp_microbench.txt): the fused form costs
//                     one issue slot where mov+fma costs two.  Row transposes and row shifts go through the team's LDS
//                     rows (one wavefront: LDS operations complete in issue order, a wave-scope fence is the only sync).
//   host emulation:   TV is an array of 16 doubles with element-wise operators, so tests/emul runs the very same
//                     source (same operation order per lane) under g++ and the CPU sanitizers.
//
// DPP hazard (VALU write of a VGPR followed within two wait states by a DPP read of it): the compiler inserts the
// s_nop for its own DPP instructions only, and it does not look into inline assembly.  Therefore EVERY DPP operation
// of these kernels is inline assembly that begins with its own `s_nop 1`; the compiler never emits a DPP instruction
// for this code.
#pragma once
#include "odef_platform.h"
#include "ek_math.h"

namespace odef {
namespace tv {

constexpr int kTeam = 16;

// leading dimension (doubles) of the team's LDS rows: >= D, == 2 (mod 4) -- even, so that rows start on 16-byte
// boundaries (ds_read/write_b128), and 4 (mod 8) dwords, so that a column access of 16 lanes is at most 2-way conflicted
__host__ __device__ constexpr int lds_ld(int D) { return D + ((2 - D % 4) + 4) % 4; }
// rows of the team's LDS image: 16 written rows + the rows a shifted read ((NB - 1) d lanes ahead) can touch, kept zero
__host__ __device__ constexpr int lds_rows(int d, int NB) { return kTeam + (NB - 1) * d; }

#ifdef ODEF_HOST_EMUL
// ------------------------------------------------------------------------------------------------ host emulation
struct TV {
  double v[kTeam];
};
struct TB {
  bool v[kTeam];
};
struct TU {  // per-lane byte offset into a record field; kOob = "this lane does not take part"
  unsigned v[kTeam];
};
constexpr unsigned kOob = 0xFFFFFFFFu;

#define ODEF_TV_BIN(op)                                                     \
  inline TV operator op(const TV& a, const TV& b) {                         \
    TV r;                                                                   \
    for (int l = 0; l < kTeam; ++l) r.v[l] = a.v[l] op b.v[l];              \
    return r;                                                               \
  }                                                                         \
  inline TV operator op(const TV& a, double b) {                            \
    TV r;                                                                   \
    for (int l = 0; l < kTeam; ++l) r.v[l] = a.v[l] op b;                   \
    return r;                                                               \
  }                                                                         \
  inline TV operator op(double a, const TV& b) {                            \
    TV r;                                                                   \
    for (int l = 0; l < kTeam; ++l) r.v[l] = a op b.v[l];                   \
    return r;                                                               \
  }
ODEF_TV_BIN(+)
ODEF_TV_BIN(-)
ODEF_TV_BIN(*)
#undef ODEF_TV_BIN
inline TV operator-(const TV& a) {
  TV r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = -a.v[l];
  return r;
}
inline TV splat(double a) {
  TV r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = a;
  return r;
}
inline TV fma(const TV& a, const TV& b, const TV& c) {
  TV r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = std::fma(a.v[l], b.v[l], c.v[l]);
  return r;
}
inline TV fma(const TV& a, double b, const TV& c) { return fma(a, splat(b), c); }
inline TV fma(double a, const TV& b, const TV& c) { return fma(splat(a), b, c); }
inline double fma(double a, double b, double c) { return std::fma(a, b, c); }
// lane r -> tab[r] for r < n, `fill` otherwise
inline TV lane_table(const double* tab, int n, double fill) {
  TV r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = l < n ? tab[l] : fill;
  return r;
}
inline TB lane_lt(int k) {
  TB r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = l < k;
  return r;
}
inline TV select(const TB& c, const TV& a, const TV& b) {
  TV r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = c.v[l] ? a.v[l] : b.v[l];
  return r;
}
inline bool any_nan(const TV& a) {
  bool n = false;
  for (int l = 0; l < kTeam; ++l) n = n || !(a.v[l] == a.v[l]);
  return n;
}
template <int K>
inline double bcast(const TV& x) { return x.v[K]; }
// acc += bcast<K>(src) * b      /      acc -= bcast<K>(src) * b
template <int K>
inline void fma_bc(TV& acc, const TV& src, const TV& b) {
  for (int l = 0; l < kTeam; ++l) acc.v[l] = std::fma(src.v[K], b.v[l], acc.v[l]);
}
template <int K>
inline void fnma_bc(TV& acc, const TV& src, const TV& b) {
  for (int l = 0; l < kTeam; ++l) acc.v[l] = std::fma(-src.v[K], b.v[l], acc.v[l]);
}
template <int K>
inline void fma_bc(TV& acc, const TV& src, double b) { fma_bc<K>(acc, src, splat(b)); }
template <int K>
inline void fnma_bc(TV& acc, const TV& src, double b) { fnma_bc<K>(acc, src, splat(b)); }
// true if the flag (0.0 / 1.0) is set in any lane of the team
inline bool team_any(const TV& flag) {
  bool a = false;
  for (int l = 0; l < kTeam; ++l) a = a || flag.v[l] != 0.0;
  return a;
}
// out[j] = bcast<K0 + j>(src), j < n
template <int K0, int n>
inline void bcast_lanes(const TV& src, double* out) {
  for (int j = 0; j < n; ++j) out[j] = src.v[K0 + j];
}
// out[j] = bcast<K>(src[j]), j < n
template <int K, int n>
inline void bcast_vec(const TV* src, double* out) {
  for (int j = 0; j < n; ++j) out[j] = src[j].v[K];
}
// Grouped broadcast-FMAs (NEG: subtract).  On the device each group of four shares one s_nop.
//   cols :  acc[c] += bcast<c>(src) * b          c = C0 .. C0+n-1   (rank-one update of the own row; pivot-column step)
//   rows :  acc[j] += bcast<K>(src[j]) * b       j = 0 .. n-1       (row K of a matrix times the lane's scalar b)
//   dot  :  acc    += sum_j bcast<K>(src[j]) * b[j]                 (forward substitution: row K of L against the lane's vector)
//   lanes:  acc    += sum_c bcast<c>(src) * b[c] c = C0 .. C0+n-1   (backward substitution / matrix-vector product)
template <bool NEG, int K>
inline void fb1(TV& acc, const TV& src, const TV& b) {
  if constexpr (NEG) fnma_bc<K>(acc, src, b);
  else fma_bc<K>(acc, src, b);
}
template <bool NEG, int C0, int n>
inline void fb_cols(TV* acc, const TV& src, const TV& b) {
  if constexpr (n > 0) {
    fb1<NEG, C0>(acc[C0], src, b);
    fb_cols<NEG, C0 + 1, n - 1>(acc, src, b);
  }
}
template <bool NEG, int K, int n>
inline void fb_rows(TV* acc, const TV* src, const TV& b) {
  for (int j = 0; j < n; ++j) fb1<NEG, K>(acc[j], src[j], b);
}
template <bool NEG, int K, int n>
inline void fb_dot(TV& acc, const TV* src, const TV* b) {
  for (int j = 0; j < n; ++j) fb1<NEG, K>(acc, src[j], b[j]);
}
template <bool NEG, int C0, int n>
inline void fb_lanes(TV& acc, const TV& src, const TV* b) {
  if constexpr (n > 0) {
    fb1<NEG, C0>(acc, src, b[C0]);
    fb_lanes<NEG, C0 + 1, n - 1>(acc, src, b);
  }
}
template <int C0, int n>
inline void fnma_bc_cols(TV* acc, const TV& src, const TV& b) { fb_cols<true, C0, n>(acc, src, b); }
// lane r -> vals[r / d] for r < d * NB, `fill` in the idle lanes
template <int d, int NB>
inline TV block_table(const double* vals, double fill) {
  TV r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = l < d * NB ? vals[l / d] : fill;
  return r;
}
// value of lane r + S (0 beyond the team)
template <int S>
inline TV shl(const TV& x) {
  TV r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = (l + S < kTeam) ? x.v[l + S] : 0.0;
  return r;
}

// the team's LDS rows
struct Lds {
  double* p;  // [lds_rows][LD]
};
inline void lds_clear(const Lds& m, int doubles) {
  for (int k = 0; k < doubles; ++k) m.p[k] = 0.0;
}
// lane r writes its row: m[r][c] = row[c]
template <int n>
inline void lds_put_row(const Lds& m, int LD, const TV (&row)[n]) {
  for (int l = 0; l < kTeam; ++l)
    for (int c = 0; c < n; ++c) m.p[l * LD + c] = row[c].v[l];
}
// lane r reads row r + S:  out[c] = m[r + S][c]
template <int S, int n>
inline void lds_get_row(const Lds& m, int LD, TV (&out)[n]) {
  for (int l = 0; l < kTeam; ++l)
    for (int c = 0; c < n; ++c) out[c].v[l] = m.p[(l + S) * LD + c];
}
// lane r reads the symmetric completion of the lower triangle: out[c] = m[max(r,c)][min(r,c)]
template <int n>
struct SymIdx {  // (device: the lane's n LDS offsets, computed once)
  void init(int) {}
};
template <int n>
inline void lds_get_sym(const Lds& m, int LD, const SymIdx<n>&, TV (&out)[n]) {
  for (int l = 0; l < kTeam; ++l)
    for (int c = 0; c < n; ++c) out[c].v[l] = (l >= c) ? m.p[l * LD + c] : m.p[c * LD + l];
}
inline void lds_sync() {}
// lane-private table in the team's LDS (constants that would otherwise sit in registers): lane r owns p[off + r * n ..]
template <int n>
inline void lds_put_private(const Lds& m, int off, const TV (&v)[n]) {
  for (int l = 0; l < kTeam; ++l)
    for (int c = 0; c < n; ++c) m.p[off + l * n + c] = v[c].v[l];
}
template <int n>
inline void lds_get_private(const Lds& m, int off, TV (&v)[n]) {
  for (int l = 0; l < kTeam; ++l)
    for (int c = 0; c < n; ++c) v[c].v[l] = m.p[off + l * n + c];
}

// ---- record fields in global memory, layout [slot][row][N]: per-lane byte offsets inside one slot -----------
// offsets of element `row_of_lane(r)` for trajectory i; lanes with row < 0 take no part
template <class F>
inline TU make_offsets(long N, long i, F&& row_of_lane) {
  TU o;
  for (int l = 0; l < kTeam; ++l) {
    const long row = row_of_lane(l);
    o.v[l] = row < 0 ? kOob : (unsigned)(((size_t)row * (size_t)N + (size_t)i) * sizeof(double));
  }
  return o;
}
struct Field {  // one save slot of one field
  double* base;
  size_t bytes;
  Field(double* b, size_t n) : base(b), bytes(n) {}
};
inline bool is_lane0() { return true; }
// 1.0 in the lanes whose value is NaN or infinite, else 0.0
inline TV nonfinite_flag(const TV& a) {
  TV r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = (std::fabs(a.v[l]) <= 1.79769313486231570815e+308) ? 0.0 : 1.0;
  return r;
}
inline void field_store(const Field& f, const TU& off, const TV& x) {
  for (int l = 0; l < kTeam; ++l)
    if (off.v[l] != kOob) *(double*)((char*)f.base + off.v[l]) = x.v[l];
}
inline TV field_load(const Field& f, const TU& off) {
  TV r;
  for (int l = 0; l < kTeam; ++l) r.v[l] = off.v[l] != kOob ? *(const double*)((const char*)f.base + off.v[l]) : 0.0;
  return r;
}
inline void field_store_uniform(const Field& f, const TU& off, double x) { field_store(f, off, splat(x)); }
// element [row][i] of a field slot laid out [rows][N], the same for every lane of the team
inline double field_load_uniform(const Field& f, size_t N, long i, int row) { return f.base[(size_t)row * N + (size_t)i]; }

#else
// ------------------------------------------------------------------------------------------------------ gfx950
using TV = double;
using TB = bool;
using TU = unsigned;
constexpr unsigned kOob = 0xFFFFFFFFu;
#define ODEF_TV_INLINE __device__ __attribute__((always_inline)) inline

ODEF_TV_INLINE int lane() { return (int)(threadIdx.x & (kTeam - 1)); }
ODEF_TV_INLINE TV splat(double a) { return a; }
ODEF_TV_INLINE double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
ODEF_TV_INLINE TV lane_table(const double* tab, int n, double fill) {  // select chain: use outside hot loops
  const int r = lane();
  double v = fill;
#pragma unroll
  for (int k = 0; k < n; ++k) v = (r == k) ? tab[k] : v;
  return v;
}
ODEF_TV_INLINE TB lane_lt(int k) { return lane() < k; }
ODEF_TV_INLINE TV select(TB c, TV a, TV b) { return c ? a : b; }
ODEF_TV_INLINE bool any_nan(TV a) { return !(a == a); }
ODEF_TV_INLINE bool is_lane0() { return lane() == 0; }
ODEF_TV_INLINE TV nonfinite_flag(TV a) { return (fabs(a) <= 1.79769313486231570815e+308) ? 0.0 : 1.0; }

template <int K>
ODEF_TV_INLINE double bcast(double x) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(K));
  return r;
}
template <int K>
ODEF_TV_INLINE void fma_bc(double& acc, double src, double b) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(b), "n"(K));
}
template <int K>
ODEF_TV_INLINE void fnma_bc(double& acc, double src, double b) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(b), "n"(K));
}
// true if the flag (0.0 / 1.0) is set in any lane of the team: one ballot, the team's 16 bits of it
ODEF_TV_INLINE bool team_any(double flag) {
  const unsigned long long b = __builtin_amdgcn_ballot_w64(flag != 0.0);
  const unsigned sh = (threadIdx.x & 48u);
  return ((b >> sh) & 0xFFFFull) != 0ull;
}
// out[j] = bcast<K0 + j>(src), j < n: up to four v_mov_b64_dpp behind one s_nop
template <int K0, int n>
ODEF_TV_INLINE void bcast_lanes(double src, double* out) {
  if constexpr (n >= 4) {
    asm("s_nop 1\n\t"
        "v_mov_b64_dpp %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %2, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %3, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
        : "v"(src), "n"(K0), "n"(K0 + 1), "n"(K0 + 2), "n"(K0 + 3));
    bcast_lanes<K0 + 4, n - 4>(src, out + 4);
  } else if constexpr (n == 3) {
    asm("s_nop 1\n\t"
        "v_mov_b64_dpp %0, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %2, %3 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2])
        : "v"(src), "n"(K0), "n"(K0 + 1), "n"(K0 + 2));
  } else if constexpr (n == 2) {
    asm("s_nop 1\n\t"
        "v_mov_b64_dpp %0, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
        : "=&v"(out[0]), "=&v"(out[1])
        : "v"(src), "n"(K0), "n"(K0 + 1));
  } else if constexpr (n == 1) {
    out[0] = bcast<K0>(src);
  }
}
// out[j] = bcast<K>(src[j]), j < n: up to four v_mov_b64_dpp behind one s_nop
template <int K, int n>
ODEF_TV_INLINE void bcast_vec(const double* src, double* out) {
  if constexpr (n >= 4) {
    asm("s_nop 1\n\t"
        "v_mov_b64_dpp %0, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %2, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %3, %7 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
        : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "n"(K));
    bcast_vec<K, n - 4>(src + 4, out + 4);
  } else if constexpr (n == 3) {
    asm("s_nop 1\n\t"
        "v_mov_b64_dpp %0, %3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %2, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2])
        : "v"(src[0]), "v"(src[1]), "v"(src[2]), "n"(K));
  } else if constexpr (n == 2) {
    asm("s_nop 1\n\t"
        "v_mov_b64_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
        : "=&v"(out[0]), "=&v"(out[1])
        : "v"(src[0]), "v"(src[1]), "n"(K));
  } else if constexpr (n == 1) {
    out[0] = bcast<K>(src[0]);
  }
}
// Grouped broadcast-FMAs (see the host section for what each computes): four v_fmac_f64_dpp behind ONE s_nop.  Inside a
// group no instruction reads through DPP a register the group writes (the accumulators are never DPP sources), so
// nothing inside needs a wait state.
#define ODEF_TV_DPPCTL " row_mask:0xf bank_mask:0xf\n\t"
#define ODEF_TV_DPPEND " row_mask:0xf bank_mask:0xf"
// cols: accumulators acc[C0..], one source, one multiplier, lanes C0..
#define ODEF_TV_FB4_COLS(SGN)                                                                          \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%4, %5 row_newbcast:%6" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %1, " SGN "%4, %5 row_newbcast:%7" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %2, " SGN "%4, %5 row_newbcast:%8" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %3, " SGN "%4, %5 row_newbcast:%9" ODEF_TV_DPPEND                                \
      : "+v"(acc[C0]), "+v"(acc[C0 + 1]), "+v"(acc[C0 + 2]), "+v"(acc[C0 + 3])                         \
      : "v"(src), "v"(b), "n"(C0), "n"(C0 + 1), "n"(C0 + 2), "n"(C0 + 3))
#define ODEF_TV_FB3_COLS(SGN)                                                                          \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%3, %4 row_newbcast:%5" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %1, " SGN "%3, %4 row_newbcast:%6" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %2, " SGN "%3, %4 row_newbcast:%7" ODEF_TV_DPPEND                                \
      : "+v"(acc[C0]), "+v"(acc[C0 + 1]), "+v"(acc[C0 + 2])                                            \
      : "v"(src), "v"(b), "n"(C0), "n"(C0 + 1), "n"(C0 + 2))
#define ODEF_TV_FB2_COLS(SGN)                                                                          \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%2, %3 row_newbcast:%4" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %1, " SGN "%2, %3 row_newbcast:%5" ODEF_TV_DPPEND                                \
      : "+v"(acc[C0]), "+v"(acc[C0 + 1])                                                               \
      : "v"(src), "v"(b), "n"(C0), "n"(C0 + 1))
// rows: accumulators acc[0..], sources src[0..], one multiplier, one lane K
#define ODEF_TV_FB4_ROWS(SGN)                                                                          \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%4, %8 row_newbcast:%9" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %1, " SGN "%5, %8 row_newbcast:%9" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %2, " SGN "%6, %8 row_newbcast:%9" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %3, " SGN "%7, %8 row_newbcast:%9" ODEF_TV_DPPEND                                \
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                         \
      : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(b), "n"(K))
#define ODEF_TV_FB3_ROWS(SGN)                                                                          \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%3, %6 row_newbcast:%7" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %1, " SGN "%4, %6 row_newbcast:%7" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %2, " SGN "%5, %6 row_newbcast:%7" ODEF_TV_DPPEND                                \
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2])                                                       \
      : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(b), "n"(K))
#define ODEF_TV_FB2_ROWS(SGN)                                                                          \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%2, %4 row_newbcast:%5" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %1, " SGN "%3, %4 row_newbcast:%5" ODEF_TV_DPPEND                                \
      : "+v"(acc[0]), "+v"(acc[1])                                                                     \
      : "v"(src[0]), "v"(src[1]), "v"(b), "n"(K))
// dot: one accumulator, sources src[0..], multipliers b[0..], one lane K
#define ODEF_TV_FB4_DOT(SGN)                                                                           \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%1, %5 row_newbcast:%9" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%2, %6 row_newbcast:%9" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%3, %7 row_newbcast:%9" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%4, %8 row_newbcast:%9" ODEF_TV_DPPEND                                \
      : "+v"(acc)                                                                                      \
      : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "n"(K))
#define ODEF_TV_FB3_DOT(SGN)                                                                           \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%1, %4 row_newbcast:%7" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%2, %5 row_newbcast:%7" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%3, %6 row_newbcast:%7" ODEF_TV_DPPEND                                \
      : "+v"(acc)                                                                                      \
      : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "n"(K))
#define ODEF_TV_FB2_DOT(SGN)                                                                           \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%1, %3 row_newbcast:%5" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%2, %4 row_newbcast:%5" ODEF_TV_DPPEND                                \
      : "+v"(acc)                                                                                      \
      : "v"(src[0]), "v"(src[1]), "v"(b[0]), "v"(b[1]), "n"(K))
// lanes: one accumulator, one source, multipliers b[C0..], lanes C0..
#define ODEF_TV_FB4_LANES(SGN)                                                                         \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%1, %2 row_newbcast:%6" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%1, %3 row_newbcast:%7" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%1, %4 row_newbcast:%8" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%1, %5 row_newbcast:%9" ODEF_TV_DPPEND                                \
      : "+v"(acc)                                                                                      \
      : "v"(src), "v"(b[C0]), "v"(b[C0 + 1]), "v"(b[C0 + 2]), "v"(b[C0 + 3]), "n"(C0), "n"(C0 + 1), "n"(C0 + 2), "n"(C0 + 3))
#define ODEF_TV_FB3_LANES(SGN)                                                                         \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%1, %2 row_newbcast:%5" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%1, %3 row_newbcast:%6" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%1, %4 row_newbcast:%7" ODEF_TV_DPPEND                                \
      : "+v"(acc)                                                                                      \
      : "v"(src), "v"(b[C0]), "v"(b[C0 + 1]), "v"(b[C0 + 2]), "n"(C0), "n"(C0 + 1), "n"(C0 + 2))
#define ODEF_TV_FB2_LANES(SGN)                                                                         \
  asm("s_nop 1\n\t"                                                                                    \
      "v_fmac_f64_dpp %0, " SGN "%1, %2 row_newbcast:%4" ODEF_TV_DPPCTL                                \
      "v_fmac_f64_dpp %0, " SGN "%1, %3 row_newbcast:%5" ODEF_TV_DPPEND                                \
      : "+v"(acc)                                                                                      \
      : "v"(src), "v"(b[C0]), "v"(b[C0 + 1]), "n"(C0), "n"(C0 + 1))
#define ODEF_TV_FB_DISPATCH(M)   \
  if constexpr (NEG) M("-");     \
  else M("")
template <bool NEG, int K>
ODEF_TV_INLINE void fb1(double& acc, double src, double b) {
  if constexpr (NEG) fnma_bc<K>(acc, src, b);
  else fma_bc<K>(acc, src, b);
}
// runs of 5 and more as ONE asm statement (one s_nop per run instead of one per group of four; generated)
#include "team_vec_groups.h"
template <bool NEG, int C0, int n>
ODEF_TV_INLINE void fb_cols(double* acc, double src, double b) {
  if constexpr (n > 12) {
    fb_cols_12<NEG, C0>(acc, src, b);
    fb_cols<NEG, C0 + 12, n - 12>(acc, src, b);
  } else if constexpr (n >= 5) {
    fb_cols_wide<NEG, C0, n>(acc, src, b);
  } else if constexpr (n >= 4) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB4_COLS);
    fb_cols<NEG, C0 + 4, n - 4>(acc, src, b);
  } else if constexpr (n == 3) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB3_COLS);
  } else if constexpr (n == 2) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB2_COLS);
  } else if constexpr (n == 1) {
    fb1<NEG, C0>(acc[C0], src, b);
  }
}
template <bool NEG, int K, int n>
ODEF_TV_INLINE void fb_rows(double* acc, const double* src, double b) {
  if constexpr (n > 14) {
    fb_rows_14<NEG, K>(acc, src, b);
    fb_rows<NEG, K, n - 14>(acc + 14, src + 14, b);
  } else if constexpr (n >= 5) {
    fb_rows_wide<NEG, K, n>(acc, src, b);
  } else if constexpr (n >= 4) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB4_ROWS);
    fb_rows<NEG, K, n - 4>(acc + 4, src + 4, b);
  } else if constexpr (n == 3) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB3_ROWS);
  } else if constexpr (n == 2) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB2_ROWS);
  } else if constexpr (n == 1) {
    fb1<NEG, K>(acc[0], src[0], b);
  }
}
template <bool NEG, int K, int n>
ODEF_TV_INLINE void fb_dot(double& acc, const double* src, const double* b) {
  if constexpr (n > 14) {
    fb_dot_14<NEG, K>(acc, src, b);
    fb_dot<NEG, K, n - 14>(acc, src + 14, b + 14);
  } else if constexpr (n >= 5) {
    fb_dot_wide<NEG, K, n>(acc, src, b);
  } else if constexpr (n >= 4) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB4_DOT);
    fb_dot<NEG, K, n - 4>(acc, src + 4, b + 4);
  } else if constexpr (n == 3) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB3_DOT);
  } else if constexpr (n == 2) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB2_DOT);
  } else if constexpr (n == 1) {
    fb1<NEG, K>(acc, src[0], b[0]);
  }
}
template <bool NEG, int C0, int n>
ODEF_TV_INLINE void fb_lanes(double& acc, double src, const double* b) {
  if constexpr (n > 14) {
    fb_lanes_14<NEG, C0>(acc, src, b);
    fb_lanes<NEG, C0 + 14, n - 14>(acc, src, b);
  } else if constexpr (n >= 5) {
    fb_lanes_wide<NEG, C0, n>(acc, src, b);
  } else if constexpr (n >= 4) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB4_LANES);
    fb_lanes<NEG, C0 + 4, n - 4>(acc, src, b);
  } else if constexpr (n == 3) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB3_LANES);
  } else if constexpr (n == 2) {
    ODEF_TV_FB_DISPATCH(ODEF_TV_FB2_LANES);
  } else if constexpr (n == 1) {
    fb1<NEG, C0>(acc, src, b[C0]);
  }
}
template <int C0, int n>
ODEF_TV_INLINE void fnma_bc_cols(double* acc, double src, double b) { fb_cols<true, C0, n>(acc, src, b); }
// lane r -> vals[r / d] for r < d * NB, `fill` in the idle lanes
template <int d, int NB>
ODEF_TV_INLINE double block_table(const double* vals, double fill) {
  const int J = lane() / d;
  double v = fill;
#pragma unroll
  for (int k = 0; k < NB; ++k) v = (J == k) ? vals[k] : v;
  return v;
}
template <int S>
ODEF_TV_INLINE double shl(double x) {  // two 32-bit row shifts (the FP64 ALU has no shifting DPP control); 0 beyond the row
  static_assert(S >= 1 && S <= 15, "row shift");
  int lo = __double2loint(x), hi = __double2hiint(x), rlo, rhi;
  asm("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shl:%4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp %1, %3 row_shl:%4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(rlo), "=&v"(rhi)
      : "v"(lo), "v"(hi), "n"(S));
  return __hiloint2double(rhi, rlo);
}

struct Lds {
  double* p;  // [lds_rows][LD], this team's slice of the workgroup's LDS
};
ODEF_TV_INLINE void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
ODEF_TV_INLINE void lds_clear(const Lds& m, int doubles) {
  for (int k = lane(); k < doubles; k += kTeam) m.p[k] = 0.0;
  lds_sync();
}
typedef double tv_double2 __attribute__((ext_vector_type(2)));
template <int n>
ODEF_TV_INLINE void lds_put_row(const Lds& m, int LD, const double (&row)[n]) {
  double* q = m.p + lane() * LD;
#pragma unroll
  for (int c = 0; c + 1 < n; c += 2) *(tv_double2*)(q + c) = tv_double2{row[c], row[c + 1]};
  if constexpr (n % 2 == 1) q[n - 1] = row[n - 1];
}
template <int S, int n>
ODEF_TV_INLINE void lds_get_row(const Lds& m, int LD, double (&out)[n]) {
  const double* q = m.p + (lane() + S) * LD;
#pragma unroll
  for (int c = 0; c + 1 < n; c += 2) {
    const tv_double2 t = *(const tv_double2*)(q + c);
    out[c] = t.x;
    out[c + 1] = t.y;
  }
  if constexpr (n % 2 == 1) out[n - 1] = q[n - 1];
}
template <int n>
ODEF_TV_INLINE void lds_put_private(const Lds& m, int off, const double (&v)[n]) {
  double* q = m.p + off + lane() * n;
#pragma unroll
  for (int c = 0; c < n; ++c) q[c] = v[c];
}
template <int n>
ODEF_TV_INLINE void lds_get_private(const Lds& m, int off, double (&v)[n]) {
  const double* q = m.p + off + lane() * n;
#pragma unroll
  for (int c = 0; c < n; ++c) v[c] = q[c];
}
template <int n>
struct SymIdx {  // the lane's offsets of m[max(r,c)][min(r,c)], c = 0..n-1: computed once per kernel
  int off[n];
  ODEF_TV_INLINE void init(int LD) {
    const int r = lane();
#pragma unroll
    for (int c = 0; c < n; ++c) off[c] = (r > c ? r : c) * LD + (r > c ? c : r);
  }
};
template <int n>
ODEF_TV_INLINE void lds_get_sym(const Lds& m, int LD, const SymIdx<n>& ix, double (&out)[n]) {
#pragma unroll
  for (int c = 0; c < n; ++c) out[c] = m.p[ix.off[c]];
}

template <class F>
ODEF_TV_INLINE unsigned make_offsets(long N, long i, F&& row_of_lane) {
  const long row = row_of_lane(lane());
  return row < 0 ? kOob : (unsigned)(((size_t)row * (size_t)N + (size_t)i) * sizeof(double));
}
// One save slot of one field as a raw buffer: lanes whose offset is kOob fall outside `bytes` and the hardware
// drops their store / returns 0 for their load -- no exec masking around the 1 + D + 1 stores of a record.
struct Field {
  __amdgpu_buffer_rsrc_t rs;
  ODEF_TV_INLINE Field(double* base, size_t bytes)
      : rs(__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000)) {}
};
typedef unsigned tv_u32x2 __attribute__((ext_vector_type(2)));
ODEF_TV_INLINE void field_store(const Field& f, unsigned off, double x) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(tv_u32x2, x), f.rs, off, 0, 0);
}
ODEF_TV_INLINE double field_load(const Field& f, unsigned off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(f.rs, off, 0, 0));
}
ODEF_TV_INLINE void field_store_uniform(const Field& f, unsigned off, double x) { field_store(f, off, x); }
ODEF_TV_INLINE double field_load_uniform(const Field& f, size_t N, long i, int row) {
  return field_load(f, (unsigned)(((size_t)row * N + (size_t)i) * sizeof(double)));
}
#endif

}  // namespace tv
}  // namespace odef
