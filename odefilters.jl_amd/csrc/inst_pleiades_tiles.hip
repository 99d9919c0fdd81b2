// One order of the register-tiled Pleiades filter (filter_tiles.h): EK0/EK1 x fixed grid/adaptive.
// Compiled five times with -DODEF_TILES_Q=1..5 (csrc/Makefile) so that the orders build in parallel.
#include "ek_kernels.h"
#ifndef ODEF_TILES_Q
#error "compile with -DODEF_TILES_Q=<order>"
#endif
#define ODEF_CAT2(a, b) a##b
#define ODEF_CAT(a, b) ODEF_CAT2(a, b)
namespace odef {
int ODEF_CAT(launch_filter_pleiades_tiles_q, ODEF_TILES_Q)(int ek1, const FilterParams& P, hipStream_t s, int adaptive) {
  LaunchTilesFilter f{P, s, adaptive};
  return dispatch_alg<RhsPleiades, ODEF_TILES_Q>(ek1, f);
}
}  // namespace odef
