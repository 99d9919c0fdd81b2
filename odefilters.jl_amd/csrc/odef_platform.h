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
using std::fabs; using std::fmax; using std::fmin; using std::sqrt; using std::pow; using std::log; using std::exp;
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
