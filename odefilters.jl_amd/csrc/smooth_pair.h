// RTS smoother, TWO lanes per trajectory (even state dimension D <= 12, large ensembles).
// src/smoothing.jl:4-63.
//
// Why.  The one-lane-per-trajectory kernel (smooth_lane.h) needs three packed D x D matrices alive per trajectory: 512
// registers per lane and 39 KB of LDS per wavefront, i.e. ONE wavefront per SIMD, which issues an FP64 instruction every
// ~8 clocks (profiles/r02_fp64_issue_rate.txt) and pays ~2 400 AGPR moves per step on top.  Here lanes 2t and 2t + 1 of a
// wavefront share trajectory t: 32 trajectories per wavefront, twice as many wavefronts, <= 256 architectural registers
// (no AGPRs), 20 KB of LDS per wavefront -- two wavefronts per SIMD.
//
// Arithmetic, in preconditioned coordinates (X = filter covariance of record s, S+ = smoothed covariance of s + 1):
//   B = A X A' + sigma2 Q = L D L'                     both lanes (replicated: a 12 x 12 factorisation does not split)
//   G = X A' B^-1                                      lane p: rows 2k + p only (two unit-triangular substitutions per row)
//   m^s = m + G (m^s_+ - A m)                          own rows
//   T = G (S+ - B)                                     own rows; S+ - B streamed from the pair's LDS image
//   S^s = X + T G'                                     entry (r, c), r >= c, by the lane that owns row r, with row c of G from
//                                                      its own registers (same parity) or from the partner (one DPP exchange
//                                                      of the six rows); = X + G (S+ - B) G', the identity test/filtering.jl:113
//                                                      asserts for the reference's stacked-QR form (src/smoothing.jl:53-57)
// The smoothed covariance of s + 1 lives in the pair's LDS image between steps (each entry written by the lane that
// produced it), the smoothed mean in the registers of the lanes that own its rows: every record is read once and
// written once, as in smooth_lane.h.
//
// Written once against a "pair value" type pr::V: on the device a plain double (cross-lane operations are
// v_mov_b32_dpp quad_perm:[1,0,3,2] and v_cndmask on the lane parity), in the host emulation (tests/emul) a pair of
// doubles with element-wise operators, so the CPU suite and the sanitizers run the same source.
#pragma once
#include "ek_lane.h"

namespace odef {
namespace pr {

#ifdef ODEF_HOST_EMUL
// ------------------------------------------------------------------------------------------------ host emulation
struct V {
  double v[2];
};
struct M {
  bool v[2];
};
#define ODEF_PR_BIN(op)                                                                                   \
  inline V operator op(const V& a, const V& b) { return V{{a.v[0] op b.v[0], a.v[1] op b.v[1]}}; }         \
  inline V operator op(const V& a, double b) { return V{{a.v[0] op b, a.v[1] op b}}; }                    \
  inline V operator op(double a, const V& b) { return V{{a op b.v[0], a op b.v[1]}}; }
ODEF_PR_BIN(+)
ODEF_PR_BIN(-)
ODEF_PR_BIN(*)
#undef ODEF_PR_BIN
inline V& operator+=(V& a, const V& b) { return a = a + b; }
inline V& operator-=(V& a, const V& b) { return a = a - b; }
inline V& operator*=(V& a, const V& b) { return a = a * b; }
inline V operator-(const V& a) { return V{{-a.v[0], -a.v[1]}}; }
inline V splat(double a) { return V{{a, a}}; }
inline V partner(const V& a) { return V{{a.v[1], a.v[0]}}; }
inline V pick(const V& a, const V& b) { return V{{a.v[0], b.v[1]}}; }  // even lane: a, odd lane: b
inline V pick(double a, double b) { return V{{a, b}}; }
inline M gt0(const V& a) { return M{{a.v[0] > 0.0, a.v[1] > 0.0}}; }
inline M is_zero(const V& a) { return M{{a.v[0] == 0.0, a.v[1] == 0.0}}; }
inline M is_nan(const V& a) { return M{{!(a.v[0] == a.v[0]), !(a.v[1] == a.v[1])}}; }
inline M operator||(const M& a, const M& b) { return M{{a.v[0] || b.v[0], a.v[1] || b.v[1]}}; }
inline bool any(const M& a) { return a.v[0] || a.v[1]; }
inline bool all(const M& a) { return a.v[0] && a.v[1]; }
inline M none() { return M{{false, false}}; }
inline V sel(const M& m, const V& a, const V& b) { return V{{m.v[0] ? a.v[0] : b.v[0], m.v[1] ? a.v[1] : b.v[1]}}; }
inline V sel(const M& m, const V& a, double b) { return sel(m, a, splat(b)); }
inline V rcp(const V& a) { return V{{1.0 / a.v[0], 1.0 / a.v[1]}}; }
inline V vsqrt(const V& a) { return V{{std::sqrt(a.v[0]), std::sqrt(a.v[1])}}; }
inline V vdiv(const V& a, const V& b) { return V{{a.v[0] / b.v[0], a.v[1] / b.v[1]}}; }
inline V ld(const double* p) { return splat(*p); }                    // both lanes of the pair read the same address
inline void st_even(double* p, const V& v) { *p = v.v[0]; }           // a value both lanes hold, stored once
inline void st_pick(double* p0, double* p1, const V& v) {             // even lane -> p0, odd lane -> p1
  *p0 = v.v[0];
  *p1 = v.v[1];
}

// Rows of one record field [row][N] of one trajectory
struct RecIn {
  const double* p;
  size_t n;
  RecIn(const double* field_row0, size_t N, size_t /*rows*/, long i, long /*i0*/) : p(field_row0 + i), n(N) {}
  V get() {
    const V r = splat(*p);
    p += n;
    return r;
  }
};
struct RecOut {
  double* p;
  size_t n, row;
  RecOut(double* field_row0, size_t N, size_t /*rows*/, long i, long /*i0*/) : p(field_row0 + i), n(N), row(0) {}
  void put_next(const V& v) { p[(row++) * n] = v.v[0]; }  // consecutive rows, a value both lanes hold
  // the even lane's value to row e0, the odd lane's to row e0 + delta (odd_only: the even lane has nothing to store)
  void put_pick(int e0, int delta, const V& v, bool odd_only = false) {
    if (!odd_only) p[(size_t)e0 * n] = v.v[0];
    p[(size_t)(e0 + delta) * n] = v.v[1];
  }
};
// The pair's LDS image: packed lower triangle of one symmetric D x D matrix
struct PairLds {
  double* base;
  V get(int r, int c) const { return splat(base[tri(r, c)]); }
  void put(int r, int c, const V& v) const { base[tri(r, c)] = v.v[0]; }
  void put_pick(int e0, int delta, const V& v, bool odd_only = false) const {
    if (!odd_only) base[e0] = v.v[0];
    base[e0 + delta] = v.v[1];
  }
};
template <class T>
inline V uval(T x) { return splat((double)x); }
inline V uval(const V& x) { return x; }
#else
// ------------------------------------------------------------------------------------------------ device (gfx950)
using V = double;
using M = bool;
__device__ inline V splat(double a) { return a; }
__device__ inline bool odd_lane() { return (threadIdx.x & 1u) != 0u; }
__device__ inline V partner(V a) {  // the value the other lane of the pair holds
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0xB1, 0xF, 0xF, true);  // quad_perm:[1,0,3,2]
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ inline V pick(V a, V b) { return odd_lane() ? b : a; }
__device__ inline M gt0(V a) { return a > 0.0; }
__device__ inline M is_zero(V a) { return a == 0.0; }
__device__ inline M is_nan(V a) { return !(a == a); }
__device__ inline bool any(M a) { return a; }
__device__ inline bool all(M a) { return a; }
__device__ inline M none() { return false; }
__device__ inline V sel(M m, V a, V b) { return m ? a : b; }
__device__ inline V rcp(V a) { return rcp_pos(a); }
__device__ inline V vsqrt(V a) { return sqrt(a); }
__device__ inline V vdiv(V a, V b) { return a / b; }
__device__ inline V ld(const double* p) { return *p; }
__device__ inline void st_even(double* p, V v) {
  if (!odd_lane()) *p = v;
}
__device__ inline void st_pick(double* p0, double* p1, V v) { *(odd_lane() ? p1 : p0) = v; }
__device__ inline V uval(double x) { return x; }

constexpr unsigned kPairOob = 0x80000000u;  // a buffer offset beyond every record: the access is dropped by the bounds check
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Rows of one record field [row][N]: `field_row0 + i0` is wave-uniform (i0 = first trajectory of the wavefront), the row
// offset a running SGPR (soffset), lane offset (i - i0) * 8 in voffset -- no per-access VALU address arithmetic.  Both
// lanes of a pair read the same 8 bytes: a wavefront reads 256 contiguous bytes per instruction.
struct RecIn {
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff, soff, step;
  __device__ RecIn(const double* field_row0, size_t N, size_t rows, long i, long i0)
      : rs(__builtin_amdgcn_make_buffer_rsrc((void*)(field_row0 + i0), 0, (int)(rows * N * sizeof(double)), 0x00020000)),
        voff((unsigned)(i - i0) * 8u), soff(0u), step((unsigned)(N * sizeof(double))) {}
  __device__ V get() {
    const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    soff += step;
    asm volatile("" : "+s"(soff));  // a running scalar offset (see RowStore, ek_lane.h)
    return __builtin_bit_cast(double, r);
  }
};
struct RecOut {
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff, voff_even, voff_odd, vpn, soff, step;
  __device__ RecOut(double* field_row0, size_t N, size_t rows, long i, long i0)
      : rs(__builtin_amdgcn_make_buffer_rsrc((void*)(field_row0 + i0), 0, (int)(rows * N * sizeof(double)), 0x00020000)),
        voff((unsigned)(i - i0) * 8u), soff(0u), step((unsigned)(N * sizeof(double))) {
    voff_even = odd_lane() ? kPairOob : voff;
    voff_odd = odd_lane() ? voff : kPairOob;
    vpn = odd_lane() ? step : 0u;
  }
  __device__ void put_next(V v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff_even, soff, ODEF_STORE_AUX);
    soff += step;
    asm volatile("" : "+s"(soff));
  }
  // the even lanes' values to row e0, the odd lanes' to row e0 + delta: two full 256-byte runs per instruction
  __device__ void put_pick(int e0, int delta, V v, bool odd_only = false) {
    const unsigned vo = __umul24(vpn, (unsigned)delta) + (odd_only ? voff_odd : voff);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, vo, (unsigned)e0 * step, ODEF_STORE_AUX);
  }
};
// The pair's LDS image: element (r, c), r >= c, of the packed lower triangle at [tri(r, c)][column of the trajectory]; 32
// trajectories per wavefront = 256 bytes per element.  The column of a trajectory is t ^ 8 for the elements of odd rows:
// when the two lanes of a pair write DIFFERENT elements (put_pick: an even and an odd row) the 16 lanes of a store group
// then cover 32 distinct banks instead of hitting 16 of them twice; reads always have both lanes on one address.
struct PairLds {
  double* base;     // the wavefront's image
  unsigned col[2];  // column of this trajectory for elements of even / odd rows
  unsigned own;     // odd lane: 32 (one element row), even lane: 0
  __device__ V get(int r, int c) const { return base[tri(r, c) * 32 + col[r & 1]]; }
  __device__ void put(int r, int c, V v) const { base[tri(r, c) * 32 + col[r & 1]] = v; }
  __device__ void put_pick(int e0, int delta, V v, bool odd_only = false) const {
    const unsigned at = (unsigned)e0 * 32u + __umul24(own, (unsigned)delta) + (odd_lane() ? col[1] : col[0]);
    if (!odd_only || odd_lane()) base[at] = v;
  }
};
#endif

// value of a table entry as a pair value (tables are doubles on fixed grids, pair values for per-trajectory steps)
template <class Tab>
__device__ inline V tabv(const Tab& tab, int k) { return uval(tab[k]); }

// In-place L D L' of a packed symmetric matrix (unit lower L below the diagonal, 1 / D_k ON the diagonal).  A
// non-positive pivot zeroes its column and gets the reciprocal 0 -- the rule of chol_packed (ek_math.h), i.e. the factor
// the reference's QR fallback yields for a positive semi-definite matrix (src/filtering.jl:38-47).  No square roots: the
// substitutions below need the reciprocal pivots only.
template <int D>
__device__ inline void ldl_packed(V (&B)[D * (D + 1) / 2]) {
  static_for<0, D>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const V piv = B[tri(k, k)];
    const M ok = gt0(piv);
    const V inv = sel(ok, rcp(sel(ok, piv, splat(1.0))), splat(0.0));
    B[tri(k, k)] = inv;
    V v[D];
#pragma unroll
    for (int i = k + 1; i < D; ++i) {
      v[i] = B[tri(i, k)];
      B[tri(i, k)] = v[i] * inv;
    }
#pragma unroll
    for (int j = k + 1; j < D; ++j)
#pragma unroll
      for (int i = j; i < D; ++i) B[tri(i, j)] -= B[tri(i, k)] * v[j];
  });
}

template <int d, int NB>
struct PairStep {
  static constexpr int D = d * NB, HR = D / 2, TRI = D * (D + 1) / 2;
  static_assert(D % 2 == 0, "the two lanes of a pair own the even and the odd rows");

  // One RTS step (src/smoothing.jl:31-63) for the records `cin`/`min_` (filter covariance / mean of time s).
  //   tab        preconditioner table of the step (precond_fill layout)
  //   ms         in: own rows of the smoothed mean of s + 1;  out: of s   (un-preconditioned, as stored)
  //   lds        in: smoothed covariance of s + 1;            out: of s   (un-preconditioned, as stored)
  //   cout, mout records of the smoothed covariance / mean of s
  template <class Tab>
  __device__ static inline void run(const PriorConsts& pc, const Tab& tab, V sigma2, RecIn& cin, RecIn& min_,
                                    const PairLds& lds, RecOut& cout, RecOut& mout, V (&ms)[HR], M& nan_seen) {
    // ---- every load of the step in flight first (src/smoothing.jl:23-24 scale afterwards)
    V X[TRI], mt[D];
#pragma unroll
    for (int k = 0; k < TRI; ++k) X[k] = cin.get();
#pragma unroll
    for (int k = 0; k < D; ++k) mt[k] = min_.get();
    ODEF_SCHED_FENCE();
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) X[tri(a, b)] = X[tri(a, b)] * tabv(tab, kTabPP + (a / d) * MAXNB + (b / d));
    // ---- own rows r = 2k + p of X:  Y = X A' (becomes G in place), and the entries of X the own results start from
    V G[HR][D], xs[HR][HR], xc[HR][HR];
    static_for<0, HR>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      V xr[D];
#pragma unroll
      for (int c = 0; c < D; ++c) xr[c] = pick(X[symidx(2 * k, c)], X[symidx(2 * k + 1, c)]);
#pragma unroll
      for (int K = 0; K < NB; ++K)
#pragma unroll
        for (int b = 0; b < d; ++b) {
          V t = xr[K * d + b];
#pragma unroll
          for (int j = K + 1; j < NB; ++j) t += pc.At[K][j] * xr[j * d + b];
          G[k][K * d + b] = t;
        }
#pragma unroll
      for (int k2 = 0; k2 <= k; ++k2) {
        xs[k][k2] = pick(xr[2 * k2], xr[2 * k2 + 1]);  // X[own row k][own row k2]
        xc[k][k2] = pick(xr[2 * k2 + 1], xr[2 * k2]);  // X[own row k][the partner's row k2]
      }
    });
    ODEF_SCHED_FENCE();
    // ---- predict (src/smoothing.jl:38), replicated in both lanes
    predict_cov_inplace<d, NB>(pc, X, sigma2);
    ODEF_SCHED_FENCE();
    // ---- M = S+ - B into the LDS image (both lanes write the same values)
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        const V sp = lds.get(a, b);
        lds.put(a, b, sp * tabv(tab, kTabPP + (a / d) * MAXNB + (b / d)) - X[tri(a, b)]);
      }
    ODEF_SCHED_FENCE();
    ldl_packed<D>(X);
    ODEF_SCHED_FENCE();
    // ---- G = Y B^-1: rows of G solve  L D L' g' = y'
    static_for<0, D>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
#pragma unroll
      for (int k = 0; k < HR; ++k) {
        V t = G[k][c];
#pragma unroll
        for (int j = 0; j < c; ++j) t -= X[tri(c, j)] * G[k][j];
        G[k][c] = t;
      }
    });
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int k = 0; k < HR; ++k) G[k][c] = G[k][c] * X[tri(c, c)];
    static_for<0, D>([&](auto cc) {
      constexpr int c = D - 1 - decltype(cc)::value;
#pragma unroll
      for (int k = 0; k < HR; ++k) {
        V t = G[k][c];
#pragma unroll
        for (int j = c + 1; j < D; ++j) t -= X[tri(j, c)] * G[k][j];
        G[k][c] = t;
      }
    });
    ODEF_SCHED_FENCE();
    // ---- mean (src/smoothing.jl:42-44): m^s = m + G (m^s_+ - A m), own rows
    {
#pragma unroll
      for (int k = 0; k < D; ++k) mt[k] = mt[k] * tabv(tab, kTabPJ + k / d);
      V dl[D];
#pragma unroll
      for (int k = 0; k < HR; ++k) {
        const V mine = ms[k], theirs = partner(ms[k]);
        dl[2 * k] = pick(mine, theirs);
        dl[2 * k + 1] = pick(theirs, mine);
      }
#pragma unroll
      for (int J = 0; J < NB; ++J)
#pragma unroll
        for (int a = 0; a < d; ++a) {
          V t = mt[J * d + a];
#pragma unroll
          for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * mt[j * d + a];
          dl[J * d + a] = dl[J * d + a] * tabv(tab, kTabPJ + J) - t;  // P m^s_+ - m^-
        }
#pragma unroll
      for (int k = 0; k < HR; ++k) {
        V t = pick(mt[2 * k], mt[2 * k + 1]);
#pragma unroll
        for (int c = 0; c < D; ++c) t += G[k][c] * dl[c];
        ms[k] = t * pick(tabv(tab, kTabPIJ + (2 * k) / d), tabv(tab, kTabPIJ + (2 * k + 1) / d));  // un-precondition (src/smoothing.jl:26)
        nan_seen = nan_seen || is_nan(ms[k]);
        mout.put_pick(2 * k, 1, ms[k]);
      }
    }
    ODEF_SCHED_FENCE();
    // ---- T = G M, own rows, M streamed from the LDS image (each entry read once, used for all own rows)
    V T[HR][D];
#pragma unroll
    for (int k = 0; k < HR; ++k)
#pragma unroll
      for (int c = 0; c < D; ++c) T[k][c] = splat(0.0);
    static_for<0, D>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int c = 0; c <= j; ++c) {
        const V mv = lds.get(j, c);
#pragma unroll
        for (int k = 0; k < HR; ++k) {
          T[k][c] += G[k][j] * mv;
          if (c != j) T[k][j] += G[k][c] * mv;
        }
      }
    });
    ODEF_SCHED_FENCE();
    // un-preconditioning factors of the own rows and of the partner's rows
    V prow[HR], pcol[HR];
#pragma unroll
    for (int k = 0; k < HR; ++k) {
      prow[k] = pick(tabv(tab, kTabPIJ + (2 * k) / d), tabv(tab, kTabPIJ + (2 * k + 1) / d));
      pcol[k] = pick(tabv(tab, kTabPIJ + (2 * k + 1) / d), tabv(tab, kTabPIJ + (2 * k) / d));
    }
    // ---- S^s[r][c] = X[r][c] + T[r] . G[c] for own rows r = 2k + p and own rows c = 2k2 + p <= r
    static_for<0, HR>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
#pragma unroll
      for (int k2 = 0; k2 <= k; ++k2) {
        V t = xs[k][k2];
#pragma unroll
        for (int c = 0; c < D; ++c) t += T[k][c] * G[k2][c];
        t = (t * prow[k]) * prow[k2];
        const int e0 = tri(2 * k, 2 * k2), e1 = tri(2 * k + 1, 2 * k2 + 1);
        cout.put_pick(e0, e1 - e0, t);
        lds.put_pick(e0, e1 - e0, t);
      }
    });
    ODEF_SCHED_FENCE();
    // ---- the partner's rows of G
#pragma unroll
    for (int k = 0; k < HR; ++k)
#pragma unroll
      for (int c = 0; c < D; ++c) G[k][c] = partner(G[k][c]);
    // ---- ... and for the partner's rows c = 2k2 + 1 - p < r.  Even lanes: k2 < k; odd lanes: k2 <= k -- the entry
    // (2k + 1, 2k) is computed by both lanes (uniform code) and stored by the odd one.
    static_for<0, HR>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
#pragma unroll
      for (int k2 = 0; k2 <= k; ++k2) {
        V t = xc[k][k2];
#pragma unroll
        for (int c = 0; c < D; ++c) t += T[k][c] * G[k2][c];
        t = (t * prow[k]) * pcol[k2];
        const int e1 = tri(2 * k + 1, 2 * k2);
        if (k2 < k) {
          const int e0 = tri(2 * k, 2 * k2 + 1);
          cout.put_pick(e0, e1 - e0, t);
          lds.put_pick(e0, e1 - e0, t);
        } else {
          cout.put_pick(e1, 0, t, true);
          lds.put_pick(e1, 0, t, true);
        }
      }
    });
  }
};

}  // namespace pr

// The whole backward pass of one trajectory (both lanes of its pair call this together).  `i`: trajectory, `i0`: first
// trajectory of the wavefront (host emulation: i0 = i), `n_hi`: wave-uniform upper bound of the record counts.
template <int d, int q, bool ADAPT>
__device__ inline void smooth_pair_traj(const SmoothParams& P, long i, long i0, const pr::PairLds& lds, long n_hi) {
  using namespace pr;
  constexpr int NB = q + 1, D = d * NB, TRI = D * (D + 1) / 2, HR = D / 2;
  using S = PairStep<d, NB>;
  const long n = ADAPT ? (long)P.nsaved[i] : P.n_save;
  const size_t N = (size_t)P.N;
  // first and last record are copied (src/smoothing.jl:11); the last one starts the carried state
  V ms[HR];
  for (int w = 0; w < 2; ++w) {
    const long s = w == 0 ? 0 : n - 1;
    V c[TRI], m[D];
#pragma unroll
    for (int k = 0; k < TRI; ++k) c[k] = ld(P.cov + ((size_t)s * TRI + k) * N + i);
#pragma unroll
    for (int k = 0; k < D; ++k) m[k] = ld(P.mean + ((size_t)s * D + k) * N + i);
    ODEF_SCHED_FENCE();
#pragma unroll
    for (int k = 0; k < TRI; ++k) st_even(P.scov + ((size_t)s * TRI + k) * N + i, c[k]);
#pragma unroll
    for (int k = 0; k < D; ++k) st_even(P.smean + ((size_t)s * D + k) * N + i, m[k]);
    if (w == 1) {
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) lds.put(a, b, c[tri(a, b)]);
#pragma unroll
      for (int k = 0; k < HR; ++k) ms[k] = pick(m[2 * k], m[2 * k + 1]);
    }
  }
  M nan_seen = none();
  for (long s = n_hi - 2; s >= 1; --s) {
    if constexpr (ADAPT) {
      if (s > n - 2) continue;  // this trajectory has fewer records: it joins at its own last one
    }
    RecOut cout(P.scov + (size_t)s * TRI * N, N, TRI, i, i0), mout(P.smean + (size_t)s * D * N, N, D, i, i0);
    bool skip;
    V sigma2;
    if constexpr (ADAPT) {
      const V h = ld(P.tsave + (size_t)(s + 1) * N + i) - ld(P.tsave + (size_t)s * N + i);
      skip = all(is_zero(h));  // (both lanes of a pair share the trajectory, hence the step)
      if (!skip) {
        sigma2 = ld(P.diff + (size_t)(s + 1) * N + i);
        V tabv_[kTabStride];
        // P(h) by the reference's running product (src/preconditioning.jl:9-14), per trajectory (as smooth_lane.h)
        V hq = splat(1.0);
#pragma unroll
        for (int k = 0; k < q; ++k) hq = hq * h;
        V val = vdiv(splat(1.0), hq * vsqrt(h));
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          tabv_[kTabPJ + J] = val;
          tabv_[kTabPIJ + J] = vdiv(splat(1.0), val);
          val = val * h;
        }
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
          for (int K = 0; K <= J; ++K) tabv_[kTabPP + J * MAXNB + K] = tabv_[kTabPJ + J] * tabv_[kTabPJ + K];
        struct {
          const V* p;
          __device__ inline V operator[](int k) const { return p[k]; }
        } tab{tabv_};
        RecIn cin(P.cov + (size_t)s * TRI * N, N, TRI, i, i0), min_(P.mean + (size_t)s * D * N, N, D, i, i0);
        S::run(P.pc, tab, sigma2, cin, min_, lds, cout, mout, ms, nan_seen);
      }
    } else {
      skip = uniform_load(P.hs + s) == 0.0;
      if (!skip) {
        sigma2 = ld(P.diff + (size_t)(s + 1) * N + i);
        const GlobalTab tab{P.ptab + (size_t)uniform_load(P.tab_idx + s) * kTabStride};
        RecIn cin(P.cov + (size_t)s * TRI * N, N, TRI, i, i0), min_(P.mean + (size_t)s * D * N, N, D, i, i0);
        S::run(P.pc, tab, sigma2, cin, min_, lds, cout, mout, ms, nan_seen);
      }
    }
    if (skip) {  // a repeated save time: the smoothed state is carried through (src/smoothing.jl:13-16)
#pragma unroll
      for (int k = 0; k < HR; ++k) mout.put_pick(2 * k, 1, ms[k]);
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) cout.put_next(lds.get(a, b));
    }
  }
  if (any(nan_seen)) P.retcode[i] = 3;  // "NaNs after smoothing" (src/smoothing.jl:25)
}

}  // namespace odef
