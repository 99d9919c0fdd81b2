// Compilation target switch for the *math* headers only.
// Product build: hipcc for gfx950 (the only shipped target).  ODEF_HOST_EMUL is defined
// solely by tests/emul/ to run the very same per-lane source under g++ (and the CPU
// sanitizers) on machines without a GPU; nothing in the library or the host mirror uses it.
#pragma once
#ifdef ODEF_HOST_EMUL
#include <cmath>
#include <cstddef>
#define __device__
#define __host__
using std::fabs; using std::fmax; using std::fmin; using std::sqrt; using std::pow; using std::log; using std::exp; using std::frexp;
#elif defined(__HIPCC_RTC__)
// hiprtc-style compilation without system headers (kept working although csrc/jit.hip now uses a hipcc child
// process): the HIP device API and the math functions are built in; the few traits the math headers use are declared here.
namespace std {
template <class T, T v>
struct integral_constant {
  static constexpr T value = v;
  using value_type = T;
  using type = integral_constant;
  constexpr operator value_type() const noexcept { return value; }
};
using true_type = integral_constant<bool, true>;
using false_type = integral_constant<bool, false>;
template <bool B, class T = void>
struct enable_if {};
template <class T>
struct enable_if<true, T> {
  using type = T;
};
template <bool B, class T = void>
using enable_if_t = typename enable_if<B, T>::type;
}  // namespace std
#ifndef INFINITY
#define INFINITY (__builtin_inf())
#endif
#else
#include <hip/hip_runtime.h>
#endif
#ifndef __HIPCC_RTC__
#include <math.h>
#include <type_traits>
#endif

// A load whose address is the same in every lane, from memory nobody writes while the kernel runs (time grid,
// preconditioner tables).  On the device it goes through the constant address space, i.e. an s_load on the scalar
// cache: counted in lgkmcnt, NOT in vmcnt.  As an ordinary (vector) global load inside a time loop that also stores
// records, its s_waitcnt vmcnt(0) waited for every record store of the previous step as well -- one full
// store round trip per step, found in the ISA of both filter kernels.
namespace odef {
#if defined(ODEF_HOST_EMUL)
template <class T>
inline T uniform_load(const T* p) { return *p; }
#else
template <class T>
__device__ __attribute__((always_inline)) inline T uniform_load(const T* p) {
  return *(const __attribute__((address_space(4))) T*)(p);
}
#endif
// Preconditioner table of a step (precond_fill, ek_math.h): in global memory, shared by all trajectories (fixed grids) ...
struct GlobalTab {
  const double* p;
  __device__ inline double operator[](int k) const { return uniform_load(p + k); }
};
// ... or in the lane's own registers (adaptive steps)
struct LocalTab {
  const double* p;
  __device__ inline double operator[](int k) const { return p[k]; }
};
}  // namespace odef
