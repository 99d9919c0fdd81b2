#include "ek_kernels.h"
namespace odef {
int launch_filter_pleiades(int q, int ek1, const TeamFilterParams& TP, hipStream_t s) {
  LaunchTeamFilter f{TP, s};
  return dispatch_order<RhsPleiades>(q, ek1, f);
}
int launch_filter_pleiades_tiles(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive) {
  LaunchTilesFilter f{P, s, adaptive};
  return dispatch_order<RhsPleiades>(q, ek1, f);
}
int launch_smooth_d28(int q, const SmoothParams& P, double* ws, hipStream_t s) {
  LaunchTeamSmooth f{P, ws, s};
  return dispatch_smooth_order<28>(q, f);
}
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
    case 1: return SmoothWs<28, 2>::size;
    case 2: return SmoothWs<28, 3>::size;
    case 3: return SmoothWs<28, 4>::size;
    case 4: return SmoothWs<28, 5>::size;
    case 5: return SmoothWs<28, 6>::size;
    default: return 0;
  }
}
}  // namespace odef
