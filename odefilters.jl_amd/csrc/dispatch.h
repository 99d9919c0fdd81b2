// Run-time (rhs_id, order, alg) -> compile-time template dispatch.
#pragma once
#include "ek_lane.h"

namespace odef {

template <class RHS, int q, class F>
inline int dispatch_alg(int ek1, F&& f) {
  if (ek1) f.template operator()<RHS, q, true>();
  else f.template operator()<RHS, q, false>();
  return 0;
}
template <class RHS, class F>
inline int dispatch_order(int q, int ek1, F&& f) {
  switch (q) {
    case 1: return dispatch_alg<RHS, 1>(ek1, f);
    case 2: return dispatch_alg<RHS, 2>(ek1, f);
    case 3: return dispatch_alg<RHS, 3>(ek1, f);
    case 4: return dispatch_alg<RHS, 4>(ek1, f);
    case 5: return dispatch_alg<RHS, 5>(ek1, f);
    default: return -2;
  }
}
// ONLYQ != 0: a translation unit built for one order (run-time compiled fields, jit.hip) instantiates that order alone
template <int d, int ONLYQ = 0, class F>
inline int dispatch_smooth_order(int q, F&& f) {
  if constexpr (ONLYQ != 0) {
    if (q != ONLYQ) return -2;
    f.template operator()<d, ONLYQ>();
    return 0;
  } else
  switch (q) {
    case 1: f.template operator()<d, 1>(); return 0;
    case 2: f.template operator()<d, 2>(); return 0;
    case 3: f.template operator()<d, 3>(); return 0;
    case 4: f.template operator()<d, 4>(); return 0;
    case 5: f.template operator()<d, 5>(); return 0;
    default: return -2;
  }
}

}  // namespace odef
