// Host-side launch paths of the workgroup-per-trajectory kernels (matrix-core filter / smoother / dense output / sampler,
// record stage), written once for any vector field with an even d <= 32 and instantiated per field (inst_pleiades.hip: d = 28,
// with the register-tiled VALU filter of round 1 as an alternate; inst_lorenz96.hip: d = 16, matrix cores only).
#pragma once
#include "ek_kernels.h"

namespace odef {

// grid.z of the staging kernels counts records: at most 65 535 per launch
inline void launch_stage_copy(bool in, const double* src, double* dst, long N, long TRI, long ld, long n_rec, hipStream_t s) {
  const dim3 tiles((unsigned)((N + kStageTile - 1) / kStageTile), (unsigned)((TRI + kStageTile - 1) / kStageTile));
  for (long r0 = 0; r0 < n_rec; r0 += 65535) {
    const unsigned nz = (unsigned)(n_rec - r0 < 65535 ? n_rec - r0 : 65535);
    const size_t so = (size_t)r0 * (size_t)N * (size_t)(in ? TRI : ld), dof = (size_t)r0 * (size_t)N * (size_t)(in ? ld : TRI);
    if (in)
      hipLaunchKernelGGL(stage_in_kernel<kStageTile>, dim3(tiles.x, tiles.y, nz), dim3(256), 0, s, src + so, dst + dof, N, TRI, ld);
    else
      hipLaunchKernelGGL(stage_out_kernel<kStageTile>, dim3(tiles.x, tiles.y, nz), dim3(256), 0, s, src + so, dst + dof, N, TRI, ld);
  }
}

// Fixed grid on the matrix-core kernel with every step saved: when `stage` holds all nsteps + 1 records the kernel writes
// its covariance records there (trajectory-major, whole lines) and one transposition pass moves them to P.cov.
// `by_order(q, ek1, P, s, adaptive)`: the field's filter launcher for one order.
template <int d, class ByOrder>
int team_filter_staged(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive, double* stage, size_t stage_doubles,
                       bool valu_kernel_selected, ByOrder&& by_order, long* staged_recs) {
  const long D = (long)d * (q + 1), TRI = D * (D + 1) / 2, ld = stage_record_ld(TRI), n_rec = P.nsteps + 1;
  if (staged_recs) *staged_recs = 0;
  if (adaptive || !P.everystep || valu_kernel_selected || !stage || (size_t)n_rec * (size_t)P.N * (size_t)ld > stage_doubles)
    return by_order(q, ek1, P, s, adaptive);
  FilterParams PS = P;
  PS.cov_stage = stage;
  PS.stage_ld = ld;
  const int rc = by_order(q, ek1, PS, s, 0);
  if (rc) return rc;
  launch_stage_copy(false, stage, P.cov, P.N, TRI, ld, n_rec, s);
  if (staged_recs) *staged_recs = n_rec;  // (the stage keeps them: a smoother pass that follows need not copy them in again)
  return 0;
}

// The smoother pass with the covariance records staged trajectory-major (record_stage.h), in blocks of as many records as
// `stage` holds, from the last record down: [records in] -> smoother launch over the block (carried state in the workspace)
// -> [records out].  `n_rec`: number of save slots in use (fixed grids: n_save; adaptive: the largest nsaved of the
// ensemble -- every trajectory joins in at the block that holds its own last record).  stage_doubles must hold at
// least two records.  filter_recs_in_stage == n_rec: the filter has left all its records in `stage` (team_filter_staged; record r
// at r N ld, stage_doubles counted from record 1) -- the pass then runs as one block on them, nothing is copied in.
// Y' = A X formed by the on-chip kernel from the packed record (d a multiple of 4: register-local, ek_kernels.h) instead of
// written by the predict kernel and read back; ODEF_SMOOTH_YFROMX=0: the hand-over through the workspace (A/B, and any other d)
template <int d>
inline bool smooth_y_from_record() {
  if (d % 4 != 0) return false;
  const char* e = getenv("ODEF_SMOOTH_YFROMX");
  return !(e && e[0] == '0');
}
template <int d, int ONLYQ = 0>
int team_smooth_staged(int q, const SmoothParams& P0, long n_rec, double* ws, double* stage, size_t stage_doubles, hipStream_t s,
                       long filter_recs_in_stage) {
  const long n = n_rec, N = P0.N;
  const long D = (long)d * (q + 1), TRI = D * (D + 1) / 2, ld = stage_record_ld(TRI);
  const size_t per_rec = (size_t)N * (size_t)ld;
  const long cap = (long)(stage_doubles / per_rec);
  const bool resident = filter_recs_in_stage == n && cap >= n - 1;
  if (resident) stage += per_rec;  // (the block starts at record 1)
  if (n < 2 || n > P0.n_save || cap < 2) return -4;  // the caller runs the pass on the records in place
  // record 0 is never smoothed (src/smoothing.jl:11) and never staged: copied here (a trajectory that has no other record
  // is not visited by any launch)
  if (hipMemcpyAsync(P0.scov, P0.cov, (size_t)TRI * (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return -5;
  if (hipMemcpyAsync(P0.smean, P0.mean, (size_t)D * (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return -5;
  long top = n - 1;  // highest record not yet through the stage
  while (top >= 1) {
    const long hi = top, lo = hi - cap + 1 > 1 ? hi - cap + 1 : 1;
    SmoothParams P = P0;
    P.stage = stage;
    P.stage_s0 = lo;
    P.stage_hi = hi;
    P.stage_ld = ld;
    if (!resident) launch_stage_copy(true, P0.cov + (size_t)lo * TRI * N, stage, N, TRI, ld, hi - lo + 1, s);
    if (!pleiades_smooth_split()) {
      LaunchTeamSmooth f{P, ws, s};
      const int rc = dispatch_smooth_order<d, ONLYQ>(q, f);
      if (rc) return rc;
    } else {
      // one kernel per phase and record: [set up the block] then, record by record from the top,
      // [begin record r: unpack, predict] -> [the rest of record r on chip: factor, sweeps, mean, G M G', the record];
      // trajectories that do not have the record (adaptive solves) or repeat a save time skip their part inside the kernels
      P.split_mode = 1;
      P.split_sc = P.split_sa = -1;
      {
        LaunchTeamSmooth f{P, ws, s};
        const int rc = dispatch_smooth_order<d, ONLYQ>(q, f);
        if (rc) return rc;
      }
      const long r_hi = hi < n - 2 ? hi : n - 2, r_lo = lo;
      P.split_mode = 2;
      for (long r = r_hi; r >= r_lo; --r) {
        P.split_sc = smooth_y_from_record<d>() ? 1 : -1;
        P.split_sa = r;
        {
          LaunchTeamSmoothPredict f{P, ws, s};
          const int rc = dispatch_smooth_order<d, ONLYQ>(q, f);
          if (rc || f.rc) return rc ? rc : f.rc;
        }
        LaunchTeamSmoothSweeps g{P, ws, s};
        const int rc = dispatch_smooth_order<d, ONLYQ>(q, g);
        if (rc || g.rc) return rc ? rc : g.rc;
      }
    }
    launch_stage_copy(false, stage, P0.scov + (size_t)lo * TRI * N, N, TRI, ld, hi - lo + 1, s);
    top = lo - 1;
  }
  return 0;
}

template <int d, int ONLYQ = 0>
int team_smooth_inplace(int q, const SmoothParams& P, double* ws, hipStream_t s) {
  LaunchTeamSmooth f{P, ws, s};
  return dispatch_smooth_order<d, ONLYQ>(q, f);
}
template <int d, int ONLYQ = 0>
int team_dense(int q, const DenseParams& P, double* ws, hipStream_t s) {
  LaunchTeamDense f{P, ws, s};
  return dispatch_smooth_order<d, ONLYQ>(q, f);
}
template <int d, int ONLYQ = 0>
int team_sample(int q, const SampleParams& P, double* ws, hipStream_t s) {
  LaunchTeamSample f{P, ws, s};
  return dispatch_smooth_order<d, ONLYQ>(q, f);
}
template <int d, int ONLYQ = 0>
size_t team_smooth_ws(int q) {
  if constexpr (ONLYQ != 0) return q == ONLYQ ? MfmaSmoothWs<d, ONLYQ + 1>::size : 0;
  else
  switch (q) {
    case 1: return MfmaSmoothWs<d, 2>::size;
    case 2: return MfmaSmoothWs<d, 3>::size;
    case 3: return MfmaSmoothWs<d, 4>::size;
    case 4: return MfmaSmoothWs<d, 5>::size;
    case 5: return MfmaSmoothWs<d, 6>::size;
    default: return 0;
  }
}

}  // namespace odef
