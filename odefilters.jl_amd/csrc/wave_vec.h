// Wavefront-local vector type for the small sequential factorisations of the tiled filter.
// The code that uses it is written once in "one value per lane" form:
//   device  (gfx950): WD is a plain double in a VGPR, cross-lane traffic is v_readlane / ds_bpermute,
//                     the section is executed by ONE wavefront (the helper wavefront of filter_tiles.h) and
//                     needs no barrier at all;
//   host emulation  : WD is an array of 64 doubles with element-wise operators, so tests/emul runs the
//                     very same algorithm (same operation order per lane) under g++.
#pragma once
#include "odef_platform.h"
#include "ek_math.h"

namespace odef {
namespace wv {

constexpr int kLanes = 64;

// LDS pointers carry their address space in the type.  As generic pointers inside an out-of-line function the
// stores would become FLAT instructions, and FLAT and DS accesses of one wavefront are not ordered against each
// other by a wave-scope fence (observed: lane 0 reading R through DS before lane k's FLAT store had landed).
// The algorithms are force-inlined as well (ODEF_WV_FN).
#ifdef ODEF_HOST_EMUL
#define ODEF_WV_INLINE inline
#define ODEF_WV_FN inline
using LdsP = double*;
using LdsCP = const double*;
inline LdsP lds(double* p) { return p; }
inline LdsCP lds(const double* p) { return p; }
#else
#define ODEF_WV_INLINE __device__ __attribute__((always_inline)) inline
#define ODEF_WV_FN __device__ __attribute__((always_inline)) inline
using LdsP = __attribute__((address_space(3))) double*;
using LdsCP = const __attribute__((address_space(3))) double*;
__device__ inline LdsP lds(double* p) { return (LdsP)p; }
__device__ inline LdsCP lds(const double* p) { return (LdsCP)p; }
#endif

#ifdef ODEF_HOST_EMUL
struct WD {
  double v[kLanes];
};
struct WB {
  bool v[kLanes];
};
#define ODEF_WV_BIN(op)                                                    \
  inline WD operator op(const WD& a, const WD& b) {                        \
    WD r;                                                                  \
    for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] op b.v[l];            \
    return r;                                                              \
  }                                                                        \
  inline WD operator op(const WD& a, double b) {                           \
    WD r;                                                                  \
    for (int l = 0; l < kLanes; ++l) r.v[l] = a.v[l] op b;                 \
    return r;                                                              \
  }                                                                        \
  inline WD operator op(double a, const WD& b) {                           \
    WD r;                                                                  \
    for (int l = 0; l < kLanes; ++l) r.v[l] = a op b.v[l];                 \
    return r;                                                              \
  }
ODEF_WV_BIN(+)
ODEF_WV_BIN(-)
ODEF_WV_BIN(*)
#undef ODEF_WV_BIN
inline WD splat(double a) {
  WD r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = a;
  return r;
}
inline WD select(const WB& c, const WD& a, const WD& b) {
  WD r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = c.v[l] ? a.v[l] : b.v[l];
  return r;
}
inline WD select(const WB& c, const WD& a, double b) { return select(c, a, splat(b)); }
inline WD select(const WB& c, double a, const WD& b) { return select(c, splat(a), b); }
inline WB lane_ge(int k) {
  WB r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = l >= k;
  return r;
}
inline WB lane_eq(int k) {
  WB r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = l == k;
  return r;
}
inline WB lane_bit(int m) {
  WB r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = (l & m) != 0;
  return r;
}
inline WB lane_range(int lo, int hi) {  // lo <= lane < hi
  WB r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = l >= lo && l < hi;
  return r;
}
inline double bcast(const WD& x, int lane) { return x.v[lane]; }
inline WD shfl_xor(const WD& x, int m) {
  WD r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = x.v[l ^ m];
  return r;
}
template <int M>
inline WD shfl_xor_c(const WD& x) { return shfl_xor(x, M); }
// lane l < n reads p[l * stride], the other lanes get 0
inline WD load(const double* p, int stride, int n) {
  WD r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = l < n ? p[l * stride] : 0.0;
  return r;
}
inline void store(double* p, int stride, const WB& mask, const WD& x) {
  for (int l = 0; l < kLanes; ++l)
    if (mask.v[l]) p[l * stride] = x.v[l];
}
inline void store_from_lane(double* p, int lane, const WD& x) { *p = x.v[lane]; }
inline void store_uniform(double* p, double x) { *p = x; }
inline double load_uniform(const double* p) { return *p; }
inline WD wlog(const WD& a) {
  WD r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = std::log(a.v[l]);
  return r;
}
inline WD wabs(const WD& a) {
  WD r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = std::fabs(a.v[l]);
  return r;
}
inline WD wrcp(const WD& a) {
  WD r;
  for (int l = 0; l < kLanes; ++l) r.v[l] = 1.0 / a.v[l];
  return r;
}
#else
using WD = double;
using WB = bool;
__device__ inline int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ inline WD splat(double a) { return a; }
__device__ inline WD select(WB c, WD a, WD b) { return c ? a : b; }
__device__ inline WB lane_ge(int k) { return lane_id() >= k; }
__device__ inline WB lane_eq(int k) { return lane_id() == k; }
__device__ inline WB lane_bit(int m) { return (lane_id() & m) != 0; }
__device__ inline WB lane_range(int lo, int hi) { return lane_id() >= lo && lane_id() < hi; }
__device__ inline double bcast(WD x, int lane) {  // lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}
__device__ inline WD shfl_xor(WD x, int m) {
  const int addr = (lane_id() ^ m) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
// Value of lane (l xor M) for a compile-time M, without the LDS crossbar: gfx950's v_permlane32_swap /
// v_permlane16_swap for M = 32 / 16 and DPP moves below that (row_ror:8, two bank-masked row shifts for 4,
// quad_perm for 2 and 1) -- VALU latency instead of a ds_bpermute round trip in each of the six dependent stages of
// the butterfly.  (Checked lane by lane against ds_bpermute: tools/lane_ops_test.hip.)
template <int M>
__device__ inline int shfl_xor_c32(int x) {
  if constexpr (M == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (lane_id() & 32) ? (int)r[0] : (int)r[1];
  } else if constexpr (M == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return (lane_id() & 16) ? (int)r[0] : (int)r[1];
  } else if constexpr (M == 8) {
    return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);  // row_ror:8
  } else if constexpr (M == 4) {
    const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);  // row_shl:4 into banks 0, 2
    return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);          // row_shr:4 into banks 1, 3
  } else if constexpr (M == 2) {
    return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
  } else {
    static_assert(M == 1, "xor mask must be a power of two below 64");
    return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
  }
}
template <int M>
__device__ inline WD shfl_xor_c(WD x) {
  return __hiloint2double(shfl_xor_c32<M>(__double2hiint(x)), shfl_xor_c32<M>(__double2loint(x)));
}
__device__ inline WD load(LdsCP p, int stride, int n) { return lane_id() < n ? p[lane_id() * stride] : 0.0; }
__device__ inline void store(LdsP p, int stride, WB mask, WD x) {
  if (mask) p[lane_id() * stride] = x;
}
__device__ inline void store_from_lane(LdsP p, int lane, WD x) {
  if (lane_id() == lane) *p = x;
}
__device__ inline void store_uniform(LdsP p, double x) {
  if (lane_id() == 0) *p = x;
}
__device__ inline double load_uniform(LdsCP p) { return *p; }
__device__ inline WD wlog(WD a) { return log(a); }
__device__ inline WD wabs(WD a) { return fabs(a); }
__device__ inline WD wrcp(WD a) { return 1.0 / a; }
#endif

// Sum over the 64 lanes of P values each (P a power of two <= 32) by a reduce-scatter butterfly:
// every halving stage exchanges only the half of the values the partner keeps, so the whole thing costs
// P - 1 + (6 - log2 P) shuffles instead of 6 P.  Afterwards result j sits in p[0] of lane owner(j).
template <int P>
struct MultiSum {
  static constexpr int log2P = (P == 1) ? 0 : (P == 2) ? 1 : (P == 4) ? 2 : (P == 8) ? 3 : (P == 16) ? 4 : 5;
  static_assert((1 << log2P) == P, "P must be a power of two <= 32");
  __host__ __device__ static constexpr int owner(int j) {  // bit t (from the top) of j selects lane bit 32 >> t
    int lane = 0;
    for (int t = 0; t < log2P; ++t)
      if (j & (1 << (log2P - 1 - t))) lane |= 32 >> t;
    return lane;
  }
  ODEF_WV_INLINE static void run(WD* p) {
    static_for<0, 6>([&](auto tt) {
      constexpr int t = decltype(tt)::value;
      constexpr int m = 32 >> t;
      if constexpr (t < log2P) {
        constexpr int half = P >> (t + 1);
        const WB up = lane_bit(m);
        _Pragma("unroll")
        for (int j = 0; j < half; ++j) {
          const WD send = select(up, p[j], p[j + half]);
          const WD keep = select(up, p[j + half], p[j]);
          p[j] = keep + shfl_xor_c<m>(send);
        }
      } else {
        p[0] = p[0] + shfl_xor_c<m>(p[0]);
      }
    });
  }
};


// z' W^-1 z for the SPD d x d matrix W (LDS, leading dimension ld): right-looking Cholesky with row i of W
// in the registers of lane i; the forward substitution L y = z rides along column by column.
// A non-positive pivot zeroes its column, as in chol_small (ek_math.h).
template <int d>
ODEF_WV_FN double chol_quadform(LdsCP WM, int ld, LdsCP z) {
  static_assert(d <= kLanes, "one lane per row");
  WD w[d];
  static_for<0, d>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    w[j] = load(WM + j, ld, d);
  });
  WD s = load(z, 1, d);
  double acc = 0.0;
  static_for<0, d>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double piv = bcast(w[k], k);
    const double rs = (piv > 0.0) ? 1.0 / sqrt(piv) : 0.0;
    const WD lk = select(lane_ge(k), w[k] * rs, 0.0);
    const double yk = bcast(s, k) * rs;
    acc += yk * yk;
    s = s - lk * yk;
    static_for<k + 1, d>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      w[j] = w[j] - lk * bcast(lk, j);
    });
  });
  return acc;
}

constexpr int next_pow2(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : n <= 16 ? 16 : 32; }

// Householder QR of the 2d x d matrix G (LDS, row-major): row i in the registers of lane i.  Per reflector
// ONE reduce-scatter butterfly yields x'x and x'g_c for all remaining columns; v'g_c follows from
// v = x - alpha e_k.  Row k of R is final once reflector k has been applied and stays in lane k, so the
// forward substitution y = R^-T z, z'S^-1 z = |y|^2 and log det S = 2 sum log|R_kk| follow without touching LDS.
// Outputs (LDS): the reflectors HV[k][k..2d), beta[0..d), y[0..d).  R itself is stored only when R_out != null.
template <int d>
ODEF_WV_FN void householder_qr_solve(LdsCP G, LdsCP z, LdsP HV, LdsP beta, LdsP y, double& zSz, double& logacc, LdsP R_out = LdsP()) {
  constexpr int d2 = 2 * d;
  static_assert(d2 <= kLanes && d <= 32, "one lane per row of G");
  WD g[d];
  static_for<0, d>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    g[c] = load(G + c, d, d2);
  });
  WD diag = splat(0.0);  // lane k: alpha_k
  WD bet = splat(0.0);   // lane k: beta_k
  static_for<0, d>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int n = d - k;
    constexpr int P = next_pow2(n);
    const WD x = select(lane_ge(k), g[k], 0.0);
    WD pr[P];
    static_for<0, P>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j < n)
        pr[j] = x * g[k + j];
      else
        pr[j] = splat(0.0);
    });
    MultiSum<P>::run(pr);
    const double nrm2 = bcast(pr[0], MultiSum<P>::owner(0));
    const double nrm = sqrt(nrm2);
    const double x0 = bcast(x, k);
    const double alpha = (x0 >= 0.0) ? -nrm : nrm;
    const double v0 = x0 - alpha;
    const double vtv = nrm2 - x0 * x0 + v0 * v0;
    const double bt = (vtv > 0.0) ? 2.0 / vtv : 0.0;
    const WD v = select(lane_eq(k), v0, x);
    store(HV + k * d2, 1, lane_range(k, d2), v);
    diag = select(lane_eq(k), alpha, diag);
    bet = select(lane_eq(k), bt, bet);
    static_for<k + 1, d>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      const double u = bcast(pr[0], MultiSum<P>::owner(c - k));
      const double gk = bcast(g[c], k);
      const double sv = bt * (u - alpha * gk);
      g[c] = g[c] - sv * v;
    });
  });
  store(beta, 1, lane_range(0, d), bet);
  if (R_out) {
    store(R_out, d + 1, lane_range(0, d), diag);
    static_for<1, d>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      store(R_out + c, d, lane_range(0, c), g[c]);  // R[lane][c] for lane < c
    });
  }
  // y = R^-T z, column-oriented: R'[r][c] = R[c][r] = g[r] of lane c; the residuals are wave-uniform values
  const WD rinv = wrcp(select(lane_range(0, d), diag, 1.0));
  double res[d];
  static_for<0, d>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    res[r] = load_uniform(z + r);
  });
  WD yv = splat(0.0);
  zSz = 0.0;
  static_for<0, d>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const double yc = res[c] * bcast(rinv, c);
    yv = select(lane_eq(c), yc, yv);
    zSz += yc * yc;
    static_for<c + 1, d>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      res[r] -= bcast(g[r], c) * yc;
    });
  });
  store(y, 1, lane_range(0, d), yv);
  WD lg[1];
  lg[0] = select(lane_range(0, d), wlog(wabs(select(lane_range(0, d), diag, 1.0))), 0.0);
  MultiSum<1>::run(lg);
  logacc = bcast(lg[0], 0);
}

}  // namespace wv
}  // namespace odef
