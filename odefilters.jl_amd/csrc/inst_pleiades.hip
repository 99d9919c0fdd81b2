#include "team_launch_impl.h"
namespace odef {
// the tiled filter is instantiated one order per translation unit (inst_pleiades_tiles.hip with -DODEF_TILES_Q=q):
// its kernels are the longest compiles of the library and used to serialise the build behind this file
int launch_filter_pleiades_tiles_q1(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
int launch_filter_pleiades_tiles_q2(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
int launch_filter_pleiades_tiles_q3(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
int launch_filter_pleiades_tiles_q4(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
int launch_filter_pleiades_tiles_q5(int ek1, const FilterParams& P, hipStream_t s, int adaptive);
static int launch_filter_pleiades_tiles_order(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive) {
  switch (q) {
    case 1: return launch_filter_pleiades_tiles_q1(ek1, P, s, adaptive);
    case 2: return launch_filter_pleiades_tiles_q2(ek1, P, s, adaptive);
    case 3: return launch_filter_pleiades_tiles_q3(ek1, P, s, adaptive);
    case 4: return launch_filter_pleiades_tiles_q4(ek1, P, s, adaptive);
    case 5: return launch_filter_pleiades_tiles_q5(ek1, P, s, adaptive);
    default: return -2;
  }
}
static int filter_pleiades(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive, double* stage, size_t stage_doubles, long* staged_recs) {
  return team_filter_staged<28>(q, ek1, P, s, adaptive, stage, stage_doubles, pleiades_filter_tiles(), launch_filter_pleiades_tiles_order, staged_recs);
}
static int smooth_pleiades(int q, const SmoothParams& P, double* ws, hipStream_t s) { return team_smooth_inplace<28>(q, P, ws, s); }
static int smooth_pleiades_staged(int q, const SmoothParams& P0, long n_rec, double* ws, double* stage, size_t stage_doubles, hipStream_t s,
                                  long filter_recs_in_stage) {
  return team_smooth_staged<28>(q, P0, n_rec, ws, stage, stage_doubles, s, filter_recs_in_stage);
}
static int dense_pleiades(int q, const DenseParams& P, double* ws, hipStream_t s) { return team_dense<28>(q, P, ws, s); }
static int sample_pleiades(int q, const SampleParams& P, double* ws, hipStream_t s) { return team_sample<28>(q, P, ws, s); }
long dense_d28_grid(long items) { return items < kDenseMfmaMaxGrid ? items : kDenseMfmaMaxGrid; }
static size_t smooth_ws_pleiades(int q) { return team_smooth_ws<28>(q); }
const TeamLaunch* team_pleiades() {  // (a function-local table: a namespace-scope constant would also be emitted for the device)
  static const TeamLaunch t = {28, filter_pleiades, smooth_pleiades, smooth_pleiades_staged, dense_pleiades, sample_pleiades, smooth_ws_pleiades};
  return &t;
}
}  // namespace odef
