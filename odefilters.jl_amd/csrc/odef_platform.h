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
using std::fabs; using std::fmax; using std::fmin; using std::sqrt; using std::pow; using std::log;
#else
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <type_traits>
