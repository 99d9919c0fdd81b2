// Host-side facts about the 16-lanes-per-trajectory kernels (rows_kernels.h): which ensemble sizes they serve and how their
// grids are sized.  A header without dependencies: the compiled-in launchers (ek_kernels.h), the launches of run-time compiled
// vector fields (api.hip) and the kernels' own translation units (incl. the ones jit.hip generates) all read it.
#pragma once
#include <cstdlib>

namespace odef {

constexpr int kRowsMaxD = 16;
// Ensemble size below which the filter uses the row-team kernels.  Cost model from tools/dpp_bench.hip: a wavefront
// alone on its SIMD issues one instruction per ~3 ns whatever it is, so the lane kernel needs ~5 us per step at any
// N <= 65 536 and the row-team kernel (I instructions per wave-step, N / 4 waves) I x 3 ns x max(1, N / 4096);
// measured crossover: profiles/r02_rows_vs_lane.jsonl.  ODEF_FILTER_ROWS_MAX_N overrides it (read at every launch,
// so tests can exercise both kernels).
constexpr long kFilterRowsMaxN = 12288;
inline long filter_rows_max_n() {
  const char* e = getenv("ODEF_FILTER_ROWS_MAX_N");
  return e ? atol(e) : kFilterRowsMaxN;
}
// Ensemble size below which the smoother of D <= 16 is the DPP row-team kernel; above: the lane kernel (N >= kSmoothLaneMinN) or the
// LDS row teams.  ODEF_SMOOTH_ROWS_MAX_N overrides the crossover (read at every launch).
constexpr long kSmoothRowsMaxN = 49152;
inline long smooth_rows_max_n() {
  const char* e = getenv("ODEF_SMOOTH_ROWS_MAX_N");
  return e ? atol(e) : kSmoothRowsMaxN;
}
// grid of those kernels: workgroups of 16 trajectories, a multiple of 8 workgroups (one contiguous trajectory range per XCD)
constexpr int kRowsWgTraj = 16;
inline unsigned rows_grid(long N) { return (unsigned)(((N + kRowsWgTraj - 1) / kRowsWgTraj + 7) / 8 * 8); }

}  // namespace odef
