// Lorenz-96 with 16 variables on the workgroup-per-trajectory kernels (state dimension 32 .. 96): the matrix-core filter (fixed
// grids and adaptive), smoother (persistent and split pass), dense output and sampler instantiated for a second shape
// (d = 16: derivative blocks of two 8-row tiles) -- the kernels of filter_mfma.h / smooth_mfma.h are not Pleiades-shaped.
#include "team_launch_impl.h"
namespace odef {
static int filter_l96_order(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive) {
  LaunchTilesFilterT<false> f{P, s, adaptive};
  return dispatch_order<RhsLorenz96>(q, ek1, f);
}
static int filter_l96(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive, double* stage, size_t stage_doubles, long* staged_recs) {
  return team_filter_staged<16>(q, ek1, P, s, adaptive, stage, stage_doubles, false, filter_l96_order, staged_recs);
}
static int smooth_l96(int q, const SmoothParams& P, double* ws, hipStream_t s) { return team_smooth_inplace<16>(q, P, ws, s); }
static int smooth_l96_staged(int q, const SmoothParams& P, long n_rec, double* ws, double* stage, size_t stage_doubles, hipStream_t s, long filter_recs_in_stage) {
  return team_smooth_staged<16>(q, P, n_rec, ws, stage, stage_doubles, s, filter_recs_in_stage);
}
static int dense_l96(int q, const DenseParams& P, double* ws, hipStream_t s) { return team_dense<16>(q, P, ws, s); }
static int sample_l96(int q, const SampleParams& P, double* ws, hipStream_t s) { return team_sample<16>(q, P, ws, s); }
static size_t smooth_ws_l96(int q) { return team_smooth_ws<16>(q); }
const TeamLaunch* team_lorenz96() {
  static const TeamLaunch t = {16, filter_l96, smooth_l96, smooth_l96_staged, dense_l96, sample_l96, smooth_ws_l96};
  return &t;
}
}  // namespace odef
