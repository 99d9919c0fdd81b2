// Internal launcher interface between the C-ABI layer (api.hip) and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include "ek_lane.h"
#include "team.h"
#include "dense_lane.h"
#include "sample_lane.h"
#include "rows_launch.h"

namespace odef {
// The launchers say which kernel they picked (printf-style; the name a profiler prints); api.hip hands it out through
// odef_kernel_name.  One slot per host thread: read back right after the launch call.
void note_kernel(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* last_kernel();
inline const char* tf(bool b) { return b ? "true" : "false"; }
// returns 0, or -2 when (rhs, q) is not instantiated
int launch_filter(int rhs, int q, int ek1, int adaptive, const FilterParams& P, hipStream_t s);
int launch_smooth(int d, int q, const SmoothParams& P, hipStream_t s);
// per-RHS translation units
int launch_filter_fhn(int q, int ek1, int adaptive, const FilterParams& P, hipStream_t s);
int launch_filter_lorenz63(int q, int ek1, int adaptive, const FilterParams& P, hipStream_t s);
int launch_filter_lotka_volterra(int q, int ek1, int adaptive, const FilterParams& P, hipStream_t s);
int launch_filter_vanderpol(int q, int ek1, int adaptive, const FilterParams& P, hipStream_t s);
int launch_filter_linear(int q, int ek1, int adaptive, const FilterParams& P, hipStream_t s);
int launch_smooth_d2(int q, const SmoothParams& P, hipStream_t s);
int launch_smooth_d3(int q, const SmoothParams& P, hipStream_t s);
// returns -3 when the state dimension is outside the dense-output kernel's range (D <= 12)
int launch_dense_d2(int q, const DenseParams& P, hipStream_t s);
int launch_dense_d3(int q, const DenseParams& P, hipStream_t s);
int launch_sample_d2(int q, const SampleParams& P, hipStream_t s);
int launch_sample_d3(int q, const SampleParams& P, hipStream_t s);
// workgroup-per-trajectory path: the launch functions of one vector field (team_launch_impl.h instantiated per field) ...
struct TeamLaunch {
  int d;
  // fixed grid (adaptive = 0; every-step records through `stage` when all of them fit) or adaptive solve on the matrix-core filter
  // `staged_recs` (may be null): set to the number of records the kernel left in `stage` (trajectory-major, record r at r N ld), 0 if none
  int (*filter)(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive, double* stage, size_t stage_doubles, long* staged_recs);
  int (*smooth)(int q, const SmoothParams& P, double* ws, hipStream_t s);  // records in place
  // `filter_recs_in_stage` == n_rec: the filter's records 0 .. n_rec - 1 are still in `stage` (nothing to copy in)
  int (*smooth_staged)(int q, const SmoothParams& P, long n_rec, double* ws, double* stage, size_t stage_doubles, hipStream_t s, long filter_recs_in_stage);
  int (*dense)(int q, const DenseParams& P, double* ws, hipStream_t s);    // ws: dense_d28_grid(items) x smooth_ws(q) doubles
  int (*sample)(int q, const SampleParams& P, double* ws, hipStream_t s);
  size_t (*smooth_ws)(int q);  // doubles of workspace per trajectory (smoother) / per grid slot (dense output, sampling)
};
// Layout stamp of what crosses between the library and a run-time compiled module of this path (jit.hip builds one from the
// headers it finds at run time: a tree whose headers moved on without a rebuild of the library must be refused, not launched)
inline unsigned long team_abi_stamp() {
  return sizeof(FilterParams) * 1000003ul + sizeof(SmoothParams) * 10007ul + sizeof(DenseParams) * 101ul + sizeof(SampleParams) + sizeof(TeamLaunch) * 7ul;
}
const TeamLaunch* team_pleiades();  // d = 28 (BASELINE config 4)
const TeamLaunch* team_lorenz96();  // d = 16: the same kernels on a second shape
const TeamLaunch* team_launch(int rhs_id);  // nullptr: the field runs on the lane / row-team kernels
long dense_d28_grid(long items);  // grid of the dense-output / sampling kernels (workspaces of smooth_ws(q) doubles)
}  // namespace odef
