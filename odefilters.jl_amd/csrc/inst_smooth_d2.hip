#include "ek_kernels.h"
namespace odef {
int launch_smooth_d2(int q, const SmoothParams& P, hipStream_t s) {
  LaunchSmooth f{P, s};
  return dispatch_smooth_order<2>(q, f);
}
int launch_dense_d2(int q, const DenseParams& P, hipStream_t s) {
  LaunchDense f{P, s};
  const int rc = dispatch_smooth_order<2>(q, f);
  return rc ? rc : f.rc;
}
int launch_sample_d2(int q, const SampleParams& P, hipStream_t s) {
  LaunchSample f{P, s};
  const int rc = dispatch_smooth_order<2>(q, f);
  return rc ? rc : f.rc;
}
}  // namespace odef
