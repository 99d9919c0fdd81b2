// Vector-field registry of libodefilter_hip (device side).
//
// The reference calls a user Julia closure `f` and its Jacobian in the middle of every
// step (src/perform_step.jl:106,116-121).  A gfx950 kernel cannot call back into the host,
// so the vector fields are compiled in and selected by `odef_rhs` (include/odefilter.h).
// Every `f` is generic in its scalar type so that the same source serves the step
// (double) and the Taylor-mode initialisation (Jet<NC>, src/state_initialization.jl:15-42).
#pragma once
#include "odef_platform.h"

namespace odef {

// ---- truncated univariate Taylor arithmetic (coefficients of (t-t0)^k) ----------------
template <int NC>
struct Jet {
  double c[NC];
  __device__ Jet() {
#pragma unroll
    for (int k = 0; k < NC; ++k) c[k] = 0.0;
  }
  __device__ Jet(double x) {
    c[0] = x;
#pragma unroll
    for (int k = 1; k < NC; ++k) c[k] = 0.0;
  }
};
template <int NC>
__device__ inline Jet<NC> operator+(const Jet<NC>& a, const Jet<NC>& b) {
  Jet<NC> r;
#pragma unroll
  for (int k = 0; k < NC; ++k) r.c[k] = a.c[k] + b.c[k];
  return r;
}
template <int NC>
__device__ inline Jet<NC> operator-(const Jet<NC>& a, const Jet<NC>& b) {
  Jet<NC> r;
#pragma unroll
  for (int k = 0; k < NC; ++k) r.c[k] = a.c[k] - b.c[k];
  return r;
}
template <int NC>
__device__ inline Jet<NC> operator-(const Jet<NC>& a) {
  Jet<NC> r;
#pragma unroll
  for (int k = 0; k < NC; ++k) r.c[k] = -a.c[k];
  return r;
}
template <int NC>
__device__ inline Jet<NC> operator*(const Jet<NC>& a, const Jet<NC>& b) {
  Jet<NC> r;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j <= k; ++j) s += a.c[j] * b.c[k - j];
    r.c[k] = s;
  }
  return r;
}
template <int NC>
__device__ inline Jet<NC> operator*(double s, const Jet<NC>& a) {
  Jet<NC> r;
#pragma unroll
  for (int k = 0; k < NC; ++k) r.c[k] = a.c[k] * s;
  return r;
}
template <int NC>
__device__ inline Jet<NC> operator*(const Jet<NC>& a, double s) { return s * a; }
template <int NC>
__device__ inline Jet<NC> operator/(const Jet<NC>& a, double s) {
  Jet<NC> r;
#pragma unroll
  for (int k = 0; k < NC; ++k) r.c[k] = a.c[k] / s;
  return r;
}
template <int NC>
__device__ inline Jet<NC> operator+(const Jet<NC>& a, double s) { Jet<NC> r = a; r.c[0] += s; return r; }
template <int NC>
__device__ inline Jet<NC> operator+(double s, const Jet<NC>& a) { return a + s; }
template <int NC>
__device__ inline Jet<NC> operator-(const Jet<NC>& a, double s) { Jet<NC> r = a; r.c[0] -= s; return r; }
template <int NC>
__device__ inline Jet<NC> operator-(double s, const Jet<NC>& a) { Jet<NC> r = -a; r.c[0] += s; return r; }

// x^a for real a:  k x0 p_k = sum_{j=1..k} (a j - (k-j)) x_j p_{k-j}
template <int NC>
__device__ inline Jet<NC> jet_pow(const Jet<NC>& x, double a) {
  Jet<NC> r;
  r.c[0] = pow(x.c[0], a);
#pragma unroll
  for (int k = 1; k < NC; ++k) {
    double s = 0.0;
#pragma unroll
    for (int j = 1; j <= k; ++j) s += (a * j - (k - j)) * x.c[j] * r.c[k - j];
    r.c[k] = s / (k * x.c[0]);
  }
  return r;
}
__device__ inline double jet_pow(double x, double a) { return pow(x, a); }
// r^-3 from r^2 (Pleiades)
__device__ inline double inv_r3(double r2) { return 1.0 / (r2 * sqrt(r2)); }
template <int NC>
__device__ inline Jet<NC> inv_r3(const Jet<NC>& r2) { return jet_pow(r2, -1.5); }

// ---- forward-mode dual numbers: the Jacobian of a vector field that provides no `jac` ----------------------
// The reference falls back to ForwardDiff when `f.jac` is missing (src/perform_step.jl:119-121): exact
// derivatives, one pass with d partials.  Same operator set as Jet.
template <int ND>
struct Dual {
  double v;
  double g[ND];
  __device__ Dual() : v(0.0) {
#pragma unroll
    for (int k = 0; k < ND; ++k) g[k] = 0.0;
  }
  __device__ Dual(double x) : v(x) {
#pragma unroll
    for (int k = 0; k < ND; ++k) g[k] = 0.0;
  }
};
#define ODEF_DUAL_LOOP for (int k = 0; k < ND; ++k)
template <int ND>
__device__ inline Dual<ND> operator+(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r;
  r.v = a.v + b.v;
#pragma unroll
  ODEF_DUAL_LOOP r.g[k] = a.g[k] + b.g[k];
  return r;
}
template <int ND>
__device__ inline Dual<ND> operator-(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r;
  r.v = a.v - b.v;
#pragma unroll
  ODEF_DUAL_LOOP r.g[k] = a.g[k] - b.g[k];
  return r;
}
template <int ND>
__device__ inline Dual<ND> operator-(const Dual<ND>& a) {
  Dual<ND> r;
  r.v = -a.v;
#pragma unroll
  ODEF_DUAL_LOOP r.g[k] = -a.g[k];
  return r;
}
template <int ND>
__device__ inline Dual<ND> operator*(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r;
  r.v = a.v * b.v;
#pragma unroll
  ODEF_DUAL_LOOP r.g[k] = a.g[k] * b.v + a.v * b.g[k];
  return r;
}
template <int ND>
__device__ inline Dual<ND> operator*(double s, const Dual<ND>& a) {
  Dual<ND> r;
  r.v = s * a.v;
#pragma unroll
  ODEF_DUAL_LOOP r.g[k] = s * a.g[k];
  return r;
}
template <int ND>
__device__ inline Dual<ND> operator*(const Dual<ND>& a, double s) { return s * a; }
template <int ND>
__device__ inline Dual<ND> operator/(const Dual<ND>& a, double s) { return (1.0 / s) * a; }
template <int ND>
__device__ inline Dual<ND> operator+(const Dual<ND>& a, double s) { Dual<ND> r = a; r.v += s; return r; }
template <int ND>
__device__ inline Dual<ND> operator+(double s, const Dual<ND>& a) { return a + s; }
template <int ND>
__device__ inline Dual<ND> operator-(const Dual<ND>& a, double s) { Dual<ND> r = a; r.v -= s; return r; }
template <int ND>
__device__ inline Dual<ND> operator-(double s, const Dual<ND>& a) { Dual<ND> r = -a; r.v += s; return r; }
template <int ND>
__device__ inline Dual<ND> jet_pow(const Dual<ND>& x, double a) {
  Dual<ND> r;
  r.v = pow(x.v, a);
  const double dv = a * pow(x.v, a - 1.0);
#pragma unroll
  ODEF_DUAL_LOOP r.g[k] = dv * x.g[k];
  return r;
}
template <int ND>
__device__ inline Dual<ND> inv_r3(const Dual<ND>& r2) { return jet_pow(r2, -1.5); }
#undef ODEF_DUAL_LOOP

template <class...>
using odef_void_t = void;
template <class RHS, class = void>
struct HasJac { static constexpr bool value = false; };
template <class RHS>
struct HasJac<RHS, odef_void_t<decltype(&RHS::jac)>> { static constexpr bool value = true; };

// J = df/du at u: the vector field's own `jac` when it has one, forward-mode differentiation of `f` otherwise
template <class RHS>
__device__ inline void rhs_jacobian(const double (&u)[RHS::d], const double* p, double (&J)[RHS::d][RHS::d]) {
  constexpr int d = RHS::d;
  if constexpr (HasJac<RHS>::value) {
    RHS::jac(u, p, J);
  } else {
    Dual<d> ud[d], fd[d];
#pragma unroll
    for (int a = 0; a < d; ++a) {
      ud[a] = Dual<d>(u[a]);
      ud[a].g[a] = 1.0;
    }
    RHS::f(ud, p, fd);
#pragma unroll
    for (int r = 0; r < d; ++r)
#pragma unroll
      for (int a = 0; a < d; ++a) J[r][a] = fd[r].g[a];
  }
}

// ---- the registry ----------------------------------------------------------------------
// Optional: a vector field may provide `team_eval_pairs` / `team_eval_assemble` (see RhsPleiades) so that
// the workgroup-per-trajectory kernels evaluate f and J with many threads instead of one.
template <class RHS, class = void>
struct HasTeamEval { static constexpr bool value = false; };
template <class RHS>
struct HasTeamEval<RHS, std::enable_if_t<RHS::has_team_eval>> { static constexpr bool value = true; };
// ... and `wave_assemble_scaled`: the assembly by one wavefront, measurement scaling included (filter_mfma.h)
template <class RHS, class = void>
struct HasWaveAssemble { static constexpr bool value = false; };
template <class RHS>
struct HasWaveAssemble<RHS, std::enable_if_t<RHS::has_wave_assemble>> { static constexpr bool value = true; };


struct RhsFHN {  // examples/fitzhughnagumo_animation.jl:8-16, README.md:36-44
  static constexpr int d = 2, np = 3, id = 0;
  static constexpr const char* name = "RhsFHN";
  template <class T>
  __device__ static void f(const T (&u)[2], const double* p, T (&du)[2]) {
    const double a = p[0], b = p[1], c = p[2];
    du[0] = c * (u[0] - u[0] * u[0] * u[0] / 3.0 + u[1]);
    du[1] = -(1.0 / c) * (u[0] - a - b * u[1]);
  }
  __device__ static void jac(const double (&u)[2], const double* p, double (&J)[2][2]) {
    const double b = p[1], c = p[2];
    J[0][0] = c * (1.0 - u[0] * u[0]);
    J[0][1] = c;
    J[1][0] = -(1.0 / c);
    J[1][1] = b / c;
  }
};

struct RhsLorenz63 {
  static constexpr int d = 3, np = 3, id = 1;
  static constexpr const char* name = "RhsLorenz63";
  template <class T>
  __device__ static void f(const T (&u)[3], const double* p, T (&du)[3]) {
    const double s = p[0], r = p[1], b = p[2];
    du[0] = s * (u[1] - u[0]);
    du[1] = u[0] * (r - u[2]) - u[1];
    du[2] = u[0] * u[1] - b * u[2];
  }
  __device__ static void jac(const double (&u)[3], const double* p, double (&J)[3][3]) {
    const double s = p[0], r = p[1], b = p[2];
    J[0][0] = -s;       J[0][1] = s;    J[0][2] = 0.0;
    J[1][0] = r - u[2]; J[1][1] = -1.0; J[1][2] = -u[0];
    J[2][0] = u[1];     J[2][1] = u[0]; J[2][2] = -b;
  }
};

struct RhsLotkaVolterra {
  static constexpr int d = 2, np = 4, id = 2;
  static constexpr const char* name = "RhsLotkaVolterra";
  template <class T>
  __device__ static void f(const T (&u)[2], const double* p, T (&du)[2]) {
    const double a = p[0], b = p[1], c = p[2], dd = p[3];
    du[0] = a * u[0] - b * u[0] * u[1];
    du[1] = -c * u[1] + dd * u[0] * u[1];
  }
  __device__ static void jac(const double (&u)[2], const double* p, double (&J)[2][2]) {
    const double a = p[0], b = p[1], c = p[2], dd = p[3];
    J[0][0] = a - b * u[1];
    J[0][1] = -b * u[0];
    J[1][0] = dd * u[1];
    J[1][1] = -c + dd * u[0];
  }
};

struct RhsVanDerPol {  // test/specific_problems.jl:44-47
  static constexpr int d = 2, np = 1, id = 3;
  static constexpr const char* name = "RhsVanDerPol";
  template <class T>
  __device__ static void f(const T (&u)[2], const double* p, T (&du)[2]) {
    const double mu = p[0];
    du[0] = u[1];
    du[1] = mu * ((1.0 - u[0] * u[0]) * u[1] - u[0]);
  }
  __device__ static void jac(const double (&u)[2], const double* p, double (&J)[2][2]) {
    const double mu = p[0];
    J[0][0] = 0.0;
    J[0][1] = 1.0;
    J[1][0] = mu * (-2.0 * u[0] * u[1] - 1.0);
    J[1][1] = mu * (1.0 - u[0] * u[0]);
  }
};

struct RhsLinear {  // test/convergence.jl:9-14, test/state_init.jl:12-17
  static constexpr int d = 2, np = 2, id = 4;
  static constexpr const char* name = "RhsLinear";
  template <class T>
  __device__ static void f(const T (&u)[2], const double* p, T (&du)[2]) {
    du[0] = p[0] * u[0];
    du[1] = p[1] * u[1];
  }
  __device__ static void jac(const double (&u)[2], const double* p, double (&J)[2][2]) {
    J[0][0] = p[0];
    J[0][1] = 0.0;
    J[1][0] = 0.0;
    J[1][1] = p[1];
  }
};


// Lorenz-96 with 16 variables, u_i' = (u_{i+1} - u_{i-2}) u_{i-1} - u_i + F, p = (F): the second vector field of the
// workgroup-per-trajectory kernels (state dimension 16 (q+1): 64 at order 3, 96 at order 5).  It has none of Pleiades' team
// evaluation hooks: the kernels' generic path evaluates f and the Jacobian in one lane.
struct RhsLorenz96 {
  static constexpr int d = 16, np = 1, id = 6;
  static constexpr const char* name = "RhsLorenz96";
  template <class T>
  __device__ static void f(const T (&u)[16], const double* p, T (&du)[16]) {
    for (int i = 0; i < 16; ++i) du[i] = (u[(i + 1) % 16] - u[(i + 14) % 16]) * u[(i + 15) % 16] - u[i] + p[0];
  }
  __device__ static void jac(const double (&u)[16], const double* /*p*/, double (&J)[16][16]) {
    for (int a = 0; a < 16; ++a)
      for (int b = 0; b < 16; ++b) J[a][b] = 0.0;
    for (int i = 0; i < 16; ++i) {
      const int ip = (i + 1) % 16, im2 = (i + 14) % 16, im1 = (i + 15) % 16;
      J[i][ip] += u[im1];
      J[i][im2] -= u[im1];
      J[i][im1] += u[ip] - u[im2];
      J[i][i] -= 1.0;
    }
  }
};

// Pleiades 7-body problem (Hairer et al. IVP test set): state (x1..7, y1..7, x'1..7, y'1..7), masses m_i = i,
//   x_i'' = sum_{j != i} m_j (x_j - x_i) / r_ij^3.   BASELINE.json config 4 (d = 28, D = 168 at order 5).
struct RhsPleiades {
  static constexpr int d = 28, np = 0, id = 5;
  static constexpr const char* name = "RhsPleiades";
  template <class T>
  __device__ static void f(const T (&u)[28], const double* /*p*/, T (&du)[28]) {
    for (int i = 0; i < 7; ++i) {
      du[i] = u[14 + i];
      du[7 + i] = u[21 + i];
    }
    for (int i = 0; i < 7; ++i) {
      T sx(0.0), sy(0.0);
      for (int j = 0; j < 7; ++j) {
        if (j == i) continue;
        const T dx = u[j] - u[i];
        const T dy = u[7 + j] - u[7 + i];
        const T w = (j + 1.0) * inv_r3(dx * dx + dy * dy);
        sx = sx + w * dx;
        sy = sy + w * dy;
      }
      du[14 + i] = sx;
      du[21 + i] = sy;
    }
  }
  __device__ static void jac(const double (&u)[28], const double* /*p*/, double (&J)[28][28]) {
    for (int a = 0; a < 28; ++a)
      for (int b = 0; b < 28; ++b) J[a][b] = 0.0;
    for (int i = 0; i < 7; ++i) {
      J[i][14 + i] = 1.0;
      J[7 + i][21 + i] = 1.0;
    }
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) {
        if (j == i) continue;
        const double mj = j + 1.0;
        const double dx = u[j] - u[i], dy = u[7 + j] - u[7 + i];
        const double r2 = dx * dx + dy * dy;
        const double r3 = 1.0 / (r2 * sqrt(r2));
        const double r5 = r3 / r2;
        const double axx = mj * (r3 - 3.0 * dx * dx * r5);
        const double axy = mj * (-3.0 * dx * dy * r5);
        const double ayy = mj * (r3 - 3.0 * dy * dy * r5);
        J[14 + i][j] += axx;      J[14 + i][i] -= axx;
        J[14 + i][7 + j] += axy;  J[14 + i][7 + i] -= axy;
        J[21 + i][j] += axy;      J[21 + i][i] -= axy;
        J[21 + i][7 + j] += ayy;  J[21 + i][7 + i] -= ayy;
      }
  }

  // Team evaluation for the workgroup-per-trajectory kernels: the 42 ordered pair interactions are
  // computed by 42 threads into `pairbuf` (5 x 49 doubles), then du (28 entries) and the raw Jacobian
  // (28 x 28, row-major) are assembled element-parallel.  Two calls with a barrier in between.
  static constexpr bool has_team_eval = true;
  static constexpr int team_scratch = 5 * 49;
  __device__ static void team_eval_pairs(int tid, const double* u, double* pairbuf) {
    if (tid < 49) {
      const int i = tid / 7, j = tid % 7;
      double axx = 0.0, axy = 0.0, ayy = 0.0, fx = 0.0, fy = 0.0;
      if (i != j) {
        const double mj = j + 1.0;
        const double dx = u[j] - u[i], dy = u[7 + j] - u[7 + i];
        const double r2 = dx * dx + dy * dy;
        const double r3 = 1.0 / (r2 * sqrt(r2));
        const double r5 = r3 / r2;
        axx = mj * (r3 - 3.0 * dx * dx * r5);
        axy = mj * (-3.0 * dx * dy * r5);
        ayy = mj * (r3 - 3.0 * dy * dy * r5);
        fx = (mj * r3) * dx;
        fy = (mj * r3) * dy;
      }
      pairbuf[0 * 49 + tid] = axx;
      pairbuf[1 * 49 + tid] = axy;
      pairbuf[2 * 49 + tid] = ayy;
      pairbuf[3 * 49 + tid] = fx;
      pairbuf[4 * 49 + tid] = fy;
    }
  }
  __device__ static void team_eval_assemble(int tid, int nthreads, const double* u, const double* pairbuf, double* du,
                                            double* Jraw /* may be null (EK0) */) {
    if (tid < 28) {
      double v;
      if (tid < 14) {
        v = u[14 + tid];
      } else {
        const int i = (tid - 14) % 7;
        const double* f = pairbuf + (tid < 21 ? 3 : 4) * 49 + i * 7;
        v = 0.0;
        for (int j = 0; j < 7; ++j) v += f[j];  // the i == j entry is zero
      }
      du[tid] = v;
    }
    if (Jraw) {
      for (int e = tid; e < 28 * 28; e += nthreads) {
        const int r = e / 28, c = e % 28;
        double v = 0.0;
        if (r < 14) {
          v = (c == r + 14) ? 1.0 : 0.0;
        } else if (c < 14) {
          const int i = (r - 14) % 7, j = c % 7;
          // block (x-acc | y-acc) x (x | y): axx, axy / axy, ayy
          const int which = (r < 21) ? (c < 7 ? 0 : 1) : (c < 7 ? 1 : 2);
          const double* a = pairbuf + which * 49 + i * 7;
          if (j != i) {
            v = a[j];
          } else {
            double sacc = 0.0;
            for (int jj = 0; jj < 7; ++jj) sacc += a[jj];
            v = -sacc;
          }
        }
        Jraw[e] = v;
      }
    }
  }
  // The same assembly for ONE wavefront that also applies the measurement scaling (filter_mfma.h, the helper's chain):
  //   du as above;  H0 = -J pi0;  M0 = H0 ql00 + I dg   (src/perform_step.jl:125, src/diffusions.jl:78), both [28][28] row-major.
  // Lane l < 56 owns half a row (row l / 2, 14 columns): the structure of J -- identity block right of the first 14 rows,
  // the three 7 x 7 interaction blocks below, zeros elsewhere -- is decided per lane, not per entry (the generic loop
  // above spends ~60 instructions per entry on index arithmetic; this one ~6).
  static constexpr bool has_wave_assemble = true;
  __device__ static void wave_assemble_scaled(int lane, const double* u, const double* pairbuf, double* du, double* H0, double* M0,
                                              double pi0, double ql00, double dg, bool ek1) {
    if (lane < 28) {
      double v;
      if (lane < 14) {
        v = u[14 + lane];
      } else {
        const int i = (lane - 14) % 7;
        const double* f = pairbuf + (lane < 21 ? 3 : 4) * 49 + i * 7;
        v = 0.0;
#pragma unroll
        for (int j = 0; j < 7; ++j) v += f[j];
      }
      du[lane] = v;
    }
    if (lane < 56) {
      const int r = lane >> 1, half = lane & 1, c0 = 14 * half;
      double jv[14];
#pragma unroll
      for (int k = 0; k < 14; ++k) jv[k] = 0.0;
      if (ek1) {
        if (r < 14) {
          if (half == 1) {
#pragma unroll
            for (int k = 0; k < 14; ++k) jv[k] = (k == r) ? 1.0 : 0.0;  // J[r][r + 14] = 1
          }
        } else if (half == 0) {
          const int i = (r - 14) % 7;
          const bool yrow = r >= 21;
          const double* ax = pairbuf + (yrow ? 1 : 0) * 49 + i * 7;  // columns 0..6:  axx (x rows) / axy (y rows)
          const double* ay = pairbuf + (yrow ? 2 : 1) * 49 + i * 7;  // columns 7..13: axy (x rows) / ayy (y rows)
          double sx = 0.0, sy = 0.0;
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            const double a = ax[j], b = ay[j];  // the j == i entries are zero
            jv[j] = a;
            jv[7 + j] = b;
            sx += a;
            sy += b;
          }
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            jv[j] = (j == i) ? -sx : jv[j];
            jv[7 + j] = (j == i) ? -sy : jv[7 + j];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 14; ++k) {
        const int c = c0 + k;
        const double h0 = (0.0 - jv[k]) * pi0;
        H0[r * 28 + c] = h0;
        M0[r * 28 + c] = h0 * ql00 + (r == c ? dg : 0.0);
      }
    }
  }
};

}  // namespace odef
