#include "ek_kernels.h"
namespace odef {
int launch_filter_lorenz63(int q, int ek1, int adaptive, const FilterParams& P, hipStream_t s) {
  LaunchFilter f{P, adaptive, s};
  return dispatch_order<RhsLorenz63>(q, ek1, f);
}
}  // namespace odef
