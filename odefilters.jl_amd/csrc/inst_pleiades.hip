#include "ek_kernels.h"
namespace odef {
int launch_filter_pleiades(int q, int ek1, const TeamFilterParams& TP, hipStream_t s) {
  LaunchTeamFilter f{TP, s};
  return dispatch_order<RhsPleiades>(q, ek1, f);
}
// the tiled filter is instantiated one order per translation unit (inst_pleiades_tiles.hip with -DODEF_TILES_Q=q):
// its kernels are the longest compiles of the library and used to serialise the build behind this file
int launch_filter_pleiades_tiles_q1(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
int launch_filter_pleiades_tiles_q2(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
int launch_filter_pleiades_tiles_q3(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
int launch_filter_pleiades_tiles_q4(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
int launch_filter_pleiades_tiles_q5(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
int launch_filter_pleiades_tiles(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive) {
  switch (q) {
    case 1: return launch_filter_pleiades_tiles_q1(ek1, P, s, adaptive);
    case 2: return launch_filter_pleiades_tiles_q2(ek1, P, s, adaptive);
    case 3: return launch_filter_pleiades_tiles_q3(ek1, P, s, adaptive);
    case 4: return launch_filter_pleiades_tiles_q4(ek1, P, s, adaptive);
    case 5: return launch_filter_pleiades_tiles_q5(ek1, P, s, adaptive);
    default: return -2;
  }
}
int launch_smooth_d28(int q, const SmoothParams& P, double* ws, hipStream_t s) {
  LaunchTeamSmooth f{P, ws, s};
  return dispatch_smooth_order<28>(q, f);
}
int launch_dense_d28(int q, const DenseParams& P, double* ws, hipStream_t s) {
  LaunchTeamDense f{P, ws, s};
  return dispatch_smooth_order<28>(q, f);
}
int launch_sample_d28(int q, const SampleParams& P, double* ws, hipStream_t s) {
  LaunchTeamSample f{P, ws, s};
  return dispatch_smooth_order<28>(q, f);
}
long dense_d28_grid(long items) { return items < kDenseMfmaMaxGrid ? items : kDenseMfmaMaxGrid; }
size_t team_filter_ws_doubles(int d, int q) {
  if (d != 28) return 0;
  switch (q) {
    case 1: return FilterWs<28, 2>::size;
    case 2: return FilterWs<28, 3>::size;
    case 3: return FilterWs<28, 4>::size;
    case 4: return FilterWs<28, 5>::size;
    case 5: return FilterWs<28, 6>::size;
    default: return 0;
  }
}
size_t team_smooth_ws_doubles(int d, int q) {
  if (d != 28) return 0;
  switch (q) {
    // the larger of the two smoothers' workspaces (smooth_team.h: 3 D x D matrices; smooth_mfma.h: 7 padded ones)
    case 1: return MfmaSmoothWs<28, 2>::size > (size_t)SmoothWs<28, 2>::size ? MfmaSmoothWs<28, 2>::size : (size_t)SmoothWs<28, 2>::size;
    case 2: return MfmaSmoothWs<28, 3>::size > (size_t)SmoothWs<28, 3>::size ? MfmaSmoothWs<28, 3>::size : (size_t)SmoothWs<28, 3>::size;
    case 3: return MfmaSmoothWs<28, 4>::size > (size_t)SmoothWs<28, 4>::size ? MfmaSmoothWs<28, 4>::size : (size_t)SmoothWs<28, 4>::size;
    case 4: return MfmaSmoothWs<28, 5>::size > (size_t)SmoothWs<28, 5>::size ? MfmaSmoothWs<28, 5>::size : (size_t)SmoothWs<28, 5>::size;
    case 5: return MfmaSmoothWs<28, 6>::size > (size_t)SmoothWs<28, 6>::size ? MfmaSmoothWs<28, 6>::size : (size_t)SmoothWs<28, 6>::size;
    default: return 0;
  }
}
}  // namespace odef
