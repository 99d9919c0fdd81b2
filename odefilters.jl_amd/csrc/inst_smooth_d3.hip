#include "ek_kernels.h"
namespace odef {
int launch_smooth_d3(int q, const SmoothParams& P, hipStream_t s) {
  LaunchSmooth f{P, s};
  return dispatch_smooth_order<3>(q, f);
}
}  // namespace odef
