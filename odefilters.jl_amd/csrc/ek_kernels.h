// Kernel templates: one lane per trajectory, one 64-lane wavefront per workgroup.
// (256 CUs x 4 SIMDs = 1024 wave slots at one wave per SIMD: 65 536 trajectories fill the
// chip exactly once; block size 64 lets the dispatcher spread waves over all SIMDs.)
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "dispatch.h"
#include "smooth_rows.h"
#include "smooth_lane.h"
#include "dense_lane.h"
#include "sample_lane.h"
#include "filter_tiles.h"
#include "record_stage.h"
#include "filter_mfma.h"
#include "rows_filter.h"
#include "rows_smooth.h"
#include "rows_kernels.h"
#include "smooth_mfma.h"
#include "smooth_onchip.h"
#include "smooth_predict.h"
#include "dense_rows.h"
#include "sample_rows.h"
#ifndef ODEF_HOST_EMUL
#include "dense_mfma.h"
#include "sample_mfma.h"
#endif
#include "launch.h"

namespace odef {

constexpr int kWave = 64;

template <class RHS, int q, bool EK1, bool EVERY, bool LAG = false>
__global__ __launch_bounds__(kWave) void ek_filter_fixed_kernel(const FilterParams P) {
  const long i0 = (long)blockIdx.x * kWave;  // wave-uniform
  // All waves run the same instruction stream and would reach their per-step store burst
  // (91 x 512 B at D = 12) together; a one-off start skew spreads the bursts over the step period
  // so that the HBM write stream is steady.  Speed only: results do not depend on it.
  if (P.stagger > 0) {
    const int n = (int)(blockIdx.x % 16u) * P.stagger;
    for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(1);  // 64 clocks each
  }
  if (i0 + threadIdx.x < P.N) filter_fixed_lane<RHS, q, EK1, EVERY, LAG>(P, i0, threadIdx.x);
}
template <class RHS, int q, bool EK1>
__global__ __launch_bounds__(kWave) void ek_filter_adaptive_kernel(const FilterParams P) {
  const long i0 = (long)blockIdx.x * kWave;
  if (i0 + threadIdx.x < P.N) filter_adaptive_lane<RHS, q, EK1>(P, i0, threadIdx.x);
}
// Smoother: row-per-lane teams (smooth_rows.h), 16 lanes per trajectory for D <= 16 (4 trajectories per
// wavefront), 32 lanes for D <= 32; per-team matrices in LDS.
template <int D>
struct SmoothTeam { static constexpr int lanes = (D <= 16) ? 16 : 32; };
template <int d, int q>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_num_vgpr(128))) void rts_smooth_kernel(const SmoothParams P) {
  constexpr int D = d * (q + 1), TEAM = SmoothTeam<D>::lanes, TPB = kWave / TEAM;
  using W = RowsWs<d, q + 1>;
  __shared__ double lds[TPB * W::size];
  const int team = threadIdx.x / TEAM, tid = threadIdx.x % TEAM;
  const long i = (long)blockIdx.x * TPB + team;
  RowState<D> st;
  if (i < P.N) smooth_rows_lane<d, q, TEAM>(P, i, tid, lds + team * W::size, &st);
}
// Smoother, one lane per trajectory (D <= 12): the read-only filter covariance of the step sits in
// lane-private LDS (78 doubles x 64 lanes = 39 KB per wave at D = 12), everything else in registers.
constexpr int kSmoothLaneMaxD = 12;
// Ensemble size from which the lane kernel is used.  Measured on Lorenz EK1(3), 1 023 steps: N = 2 048 / 4 096:
// 27.6 / 27.7 ms (lane) against 13.9 / 15.5 ms (row teams); N = 16 384: 30.4 against 42.4 ms.
// ODEF_SMOOTH_LANE_MIN_N overrides it (read at every launch, so tests can exercise both kernels).
// Ensemble size below which the every-step filter stores its records lagged by one step (LaggedSink, ek_lane.h);
// ODEF_FILTER_LAG_MAX_N overrides it (read at every launch).
constexpr long kFilterLagMaxN = 32768;
inline long filter_lag_max_n() {
  const char* e = getenv("ODEF_FILTER_LAG_MAX_N");
  return e ? atol(e) : kFilterLagMaxN;
}
constexpr long kSmoothLaneMinN = 6144;
inline long smooth_lane_min_n() {
  const char* e = getenv("ODEF_SMOOTH_LANE_MIN_N");
  return e ? atol(e) : kSmoothLaneMinN;
}
// Two kernels (fixed grid / adaptive records) so that each gets its own register allocation.
template <int d, int q, bool ADAPT>
__global__ __launch_bounds__(kWave) void rts_smooth_lane_kernel(const SmoothParams P) {
  constexpr int D = d * (q + 1), TRI = D * (D + 1) / 2;
  __shared__ double lds[TRI * kWave];
  const long i0 = (long)blockIdx.x * kWave;
  const LaneMem xl{lds + threadIdx.x, kWave};
  const bool valid = i0 + threadIdx.x < P.N;
  long n_hi = P.n_save;
  if constexpr (ADAPT) n_hi = wave_uniform_max(valid ? (long)P.nsaved[i0 + threadIdx.x] : 0, valid);
  if (valid) smooth_lane_v2<d, q, ADAPT>(P, i0, threadIdx.x, xl, n_hi);
}

// Dense output: blockIdx.y = query time, one lane per trajectory (D <= 12).
template <int d, int q>
__global__ __launch_bounds__(kWave) void dense_output_kernel(const DenseParams P) {
  constexpr int D = d * (q + 1), TRI = D * (D + 1) / 2;
  __shared__ double lds[TRI * kWave];
  const long i = (long)blockIdx.x * kWave + threadIdx.x;
  const LaneMem xl{lds + threadIdx.x, kWave};
  if (i < P.N) dense_lane<d, q>(P, i, (long)blockIdx.y, xl);
}
// ... and 12 < D <= 32: one row-per-lane team per (trajectory, query time) item (dense_rows.h)
template <int d, int q>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_num_vgpr(128))) void dense_rows_kernel(const DenseParams P) {
  constexpr int D = d * (q + 1), TEAM = SmoothTeam<D>::lanes, TPB = kWave / TEAM;
  using W = RowsWs<d, q + 1>;
  __shared__ double lds[TPB * W::size];
  const int team = threadIdx.x / TEAM, tid = threadIdx.x % TEAM;
  const long it = (long)blockIdx.x * TPB + team;  // item = (query time, trajectory), trajectory fastest
  RowState<D> st;
  if (it < P.N * P.n_q) dense_rows_lane<d, q, TEAM>(P, it % P.N, it / P.N, tid, lds + team * W::size, &st);
}
struct LaunchDense {
  const DenseParams& P;
  hipStream_t s;
  int rc = 0;
  template <int d, int q>
  void operator()() {
    if constexpr (d * (q + 1) <= kSmoothLaneMaxD) {
      dim3 grid((unsigned)((P.N + kWave - 1) / kWave), (unsigned)P.n_q);
      hipLaunchKernelGGL((dense_output_kernel<d, q>), grid, dim3(kWave), 0, s, P);
    } else if constexpr (d * (q + 1) <= 32) {
      constexpr int TPB = kWave / SmoothTeam<d*(q + 1)>::lanes;
      const long items = P.N * P.n_q;
      hipLaunchKernelGGL((dense_rows_kernel<d, q>), dim3((unsigned)((items + TPB - 1) / TPB)), dim3(kWave), 0, s, P);
    } else {
      rc = -3;
    }
  }
};

// Posterior sampling (sample_lane.h): one lane per (trajectory, sample); blockIdx.y = sample.
template <int d, int q>
__global__ __launch_bounds__(kWave) void sample_kernel(const SampleParams P) {
  constexpr int D = d * (q + 1), TRI = D * (D + 1) / 2;
  __shared__ double lds[TRI * kWave];
  const long i = (long)blockIdx.x * kWave + threadIdx.x;
  const LaneMem xl{lds + threadIdx.x, kWave};
  const bool valid = i < P.N;
  const long n_hi = (P.adaptive && !P.tq) ? wave_uniform_max(valid ? (long)P.nsaved[i] : 0, valid) : P.n_save;
  if (valid) sample_lane<d, q>(P, i, (long)blockIdx.y, xl, n_hi);
}
// ... and 12 < D <= 32: one row-per-lane team per (trajectory, sample) item (sample_rows.h)
template <int d, int q>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_num_vgpr(128))) void sample_rows_kernel(const SampleParams P) {
  constexpr int D = d * (q + 1), TEAM = SmoothTeam<D>::lanes, TPB = kWave / TEAM;
  using W = RowsWs<d, q + 1>;
  __shared__ double lds[TPB * W::size];
  const int team = threadIdx.x / TEAM, tid = threadIdx.x % TEAM;
  const long it = (long)blockIdx.x * TPB + team;  // item = (sample, trajectory), trajectory fastest
  RowState<D> st;
  if (it < P.N * P.n_samples) sample_rows_lane<d, q, TEAM>(P, it % P.N, it / P.N, tid, lds + team * W::size, &st);
}
struct LaunchSample {
  const SampleParams& P;
  hipStream_t s;
  int rc = 0;
  template <int d, int q>
  void operator()() {
    if constexpr (d * (q + 1) <= kSmoothLaneMaxD) {
      const dim3 grid((unsigned)((P.N + kWave - 1) / kWave), (unsigned)P.n_samples);
      hipLaunchKernelGGL((sample_kernel<d, q>), grid, dim3(kWave), 0, s, P);
    } else if constexpr (d * (q + 1) <= 32) {
      constexpr int TPB = kWave / SmoothTeam<d*(q + 1)>::lanes;
      const long items = P.N * P.n_samples;
      hipLaunchKernelGGL((sample_rows_kernel<d, q>), dim3((unsigned)((items + TPB - 1) / TPB)), dim3(kWave), 0, s, P);
    } else {
      rc = -3;
    }
  }
};

constexpr int kTeamBig = 256;  // threads of the workgroup-per-trajectory smoother / dense output / sampler kernels

// Register-tiled workgroup-per-trajectory filter (filter_tiles.h): 320 threads with one 7 x 7 covariance tile each
// plus one helper wavefront for the small sequential factorisations.
template <class RHS, int q, bool EK1>
__global__ __launch_bounds__(kTilesBlock) void ek_filter_tiles_kernel(const FilterParams P) {
  using TF = TilesFilter<RHS, q, EK1>;
  __shared__ double sm[TF::W::size];
  TileState st;
  const long i = team_traj(P.N);
  if (i < 0) return;
  if (threadIdx.x >= kTilesThreads)  // the helper wavefront: same barriers, its own code path
    TF::template run<true>(P, i, (int)threadIdx.x, sm, &st);
  else
    TF::template run<false>(P, i, (int)threadIdx.x, sm, &st);
}
template <class RHS, int q, bool EK1>
__global__ __launch_bounds__(kTilesBlock) void ek_filter_tiles_adaptive_kernel(const FilterParams P) {
  using TF = TilesFilter<RHS, q, EK1>;
  __shared__ double sm[TF::W::size];
  TileState st;
  const long i = team_traj(P.N);
  if (i < 0) return;
  if (threadIdx.x >= kTilesThreads)
    TF::template run_adaptive<true>(P, i, (int)threadIdx.x, sm, &st);
  else
    TF::template run_adaptive<false>(P, i, (int)threadIdx.x, sm, &st);
}
// ODEF_PLEIADES_FILTER=tiles selects the register-tiled VALU kernels (default: the MFMA kernels of filter_mfma.h, fixed
// grids and adaptive)
inline bool pleiades_filter_tiles() {
  const char* e = getenv("ODEF_PLEIADES_FILTER");
  return e && e[0] == 't';
}
// VALU_ALTERNATES: the register-tiled VALU kernels of round 1 are instantiated beside the matrix-core ones (Pleiades only: their
// 7 x 7 tiles are cut for d = 28); every other workgroup-per-trajectory field gets the matrix-core kernels alone
template <bool VALU_ALTERNATES = true>
struct LaunchTilesFilterT {
  const FilterParams& P;
  hipStream_t s;
  int adaptive = 0;
  template <class RHS, int q, bool EK1>
  void operator()() {
    if (!VALU_ALTERNATES || !pleiades_filter_tiles()) {
      note_kernel("odef::ek_filter_mfma%s_kernel<odef::%s, %d, %s>", adaptive ? "_adaptive" : "", RHS::name, q, tf(EK1));
      if (adaptive)
        hipLaunchKernelGGL((ek_filter_mfma_adaptive_kernel<RHS, q, EK1>), dim3(team_grid(P.N)), dim3(kMfBlock), 0, s, P);
      else
        hipLaunchKernelGGL((ek_filter_mfma_kernel<RHS, q, EK1>), dim3(team_grid(P.N)), dim3(kMfBlock), 0, s, P);
      return;
    }
    if constexpr (VALU_ALTERNATES) {
      note_kernel("odef::ek_filter_tiles%s_kernel<odef::%s, %d, %s>", adaptive ? "_adaptive" : "", RHS::name, q, tf(EK1));
      if (adaptive)
        hipLaunchKernelGGL((ek_filter_tiles_adaptive_kernel<RHS, q, EK1>), dim3(team_grid(P.N)), dim3(kTilesBlock), 0, s, P);
      else
        hipLaunchKernelGGL((ek_filter_tiles_kernel<RHS, q, EK1>), dim3(team_grid(P.N)), dim3(kTilesBlock), 0, s, P);
    }
  }
};
using LaunchTilesFilter = LaunchTilesFilterT<true>;

// The same pass on the matrix cores (smooth_mfma.h): 4 wavefronts per trajectory, matrices in a global workspace.
// Four workgroups per CU (128 registers): the phases are bound by the latency and traffic of the global workspace, and more
// resident workgroups hide more of it -- 319 / 307 / 273 ms with 2 / 3 / 4 (2 048 trajectories x 64 steps).
template <int d, int q, bool SPLITK = false>
__global__ __launch_bounds__(kTeamBig, 4) void rts_smooth_mfma_kernel(const SmoothParams P, double* ws) {
  using W = MfmaSmoothWs<d, q + 1>;
  __shared__ double lds[W::lds_size];
  const long i = team_traj(P.N);
  if (i < 0) return;
  smooth_mfma_traj<d, q, SPLITK>(P, i, ws + (size_t)i * W::size, lds);
}
#ifdef ODEF_SWEEPS_STAMPS  // diagnostic build (tools/split_smooth_stamps.hip): wall-clock ticks per phase of workgroup 0
__device__ unsigned long long g_sweeps_stamps[16];
__device__ unsigned long long g_sweeps_t0;
#define ODEF_SSTAMP(k)                                                           \
  do {                                                                           \
    __syncthreads();                                                             \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                   \
      const unsigned long long now_ = wall_clock64();                            \
      if ((k) >= 0) g_sweeps_stamps[(k) < 0 ? 0 : (k)] += now_ - g_sweeps_t0;    \
      g_sweeps_t0 = now_;                                                        \
    }                                                                            \
  } while (0)
#else
#define ODEF_SSTAMP(k)
#endif
// ONE record of the smoother for every trajectory, on chip (split pass, the default of the staged smoother; behind
// rts_smooth_predict_kernel, which leaves B, Y' = A X, M and the vectors in the workspace): one workgroup of DPB wavefronts per
// trajectory.  The upper tiles of B go to LDS (rows padded to 17 doubles so that the transposed reads of the backward sweep are
// bank-conflict free) and are factorised there; wavefront c holds tile column c of the right-hand sides in its accumulators
// for both sweeps -- no barrier, no re-read, the factor never leaves the chip -- and keeps G' there for the mean, for
// R = G M G' (smooth_onchip.h) and for the smoothed record.  Prototypes and measurements: tools/onchip_sweep_proto.hip,
// tools/onchip_products_proto.hip; phase stamps: tools/split_smooth_stamps.hip.
template <int d, int q>
__global__ __launch_bounds__((64 * MfmaSmoothWs<d, q + 1>::DPB)) void rts_smooth_sweeps_kernel(const SmoothParams P, double* ws) {
  using W = MfmaSmoothWs<d, q + 1>;
  constexpr int DPB = W::DPB, DP = W::DP, LDT = 17, TSZ = mf::kB * LDT;
  // k-steps (of 4 rows) of the last tile row that hold state components: the rows of Y' / G' behind the state dimension are zero,
  // the sweeps and the products skip them
  constexpr int KL = (W::D - 16 * (DPB - 1) + 3) / 4;
  extern __shared__ double lds[];
  const long i = team_traj(P.N);
  if (i < 0) return;
  double* my = ws + (size_t)i * W::size;
  if ((long)my[W::FLG] != P.split_sa) return;  // (workgroup-uniform) no factor was prepared for this record
  const int tid = (int)threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const double* BM = my + W::BM;
  const double* YT = my + W::YT;
  auto tix = [](int j, int jp) { return j * DPB - j * (j - 1) / 2 + (jp - j); };
  ODEF_SSTAMP(-1);
  // B -> LDS, upper tiles in row order, wavefront w takes the tiles w, w + DPB, ...: all its loads in flight, then the stores.
  // Behind them (loads return in order) the right-hand sides: tile column `wave` of Y' = A X goes to the accumulators, where
  // it stays until the record is done; those loads complete beside the factorisation.
  const int c0 = wave * mf::kB;
  mf::d4 acc[DPB];
  // P.split_sc == 1 (d a multiple of 4): Y' = A X is not in the workspace -- this kernel forms its tile column from the packed
  // record itself, see below
  const bool yfromx = (d % 4 == 0) && P.split_sc == 1;
  const double* rec_x = P.stage + ((size_t)(P.split_sa - P.stage_s0) * (size_t)P.N + (size_t)i) * (size_t)P.stage_ld;
  double* pj_early = lds + oc::Products<DPB>::size + 2 * DP;  // (behind everything the factor uses; P by state component)
  if (yfromx)
    for (int k = tid; k < DP; k += (int)blockDim.x) pj_early[k] = my[W::PJV + k];
  {
    constexpr int NTU = DPB * (DPB + 1) / 2, PER = (NTU + DPB - 1) / DPB;
    mf::d4 x[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = wave + u * DPB;
      int j = 0, rest = t;
      while (rest >= DPB - j) {  // block row j from t by counting down the row lengths (wavefront-uniform)
        rest -= DPB - j;
        ++j;
      }
      if (t < NTU) x[u] = oc::load_tile_major(BM + W::tile_at(j, j + rest));
    }
    if (!yfromx) {
#pragma unroll
      for (int j = 0; j < DPB; ++j) acc[j] = oc::load_tile_major(YT + W::tile_at(j, wave));
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = wave + u * DPB;
      if (t < NTU) {
#pragma unroll
        for (int v = 0; v < 4; ++v) lds[t * TSZ + (4 * v + (l >> 4)) * LDT + (l & 15)] = x[u][v];
      }
    }
  }
  __syncthreads();
  if (yfromx) {
    // X[:, c] out of the packed lower triangle of the record (element (r, col) at hi (hi + 1) / 2 + lo), unscaled: in flight while
    // the factorisation runs
#pragma unroll
    for (int t = 0; t < DPB; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = t * mf::kB + 4 * v + (l >> 4), col = c0 + (l & 15);
        const int hi = r > col ? r : col, lo = r > col ? col : r;
        acc[t][v] = hi < W::D ? rec_x[hi * (hi + 1) / 2 + lo] : 0.0;
      }
  }
  ODEF_SSTAMP(0);  // B -> LDS
  // B = U'U in LDS, right-looking by block rows: the diagonal tile is factorised by wavefront 0 and replaced by
  // W_j = L_jj^-1 (what the sweeps multiply with), the tiles of block row j become U[j, .] = W_j (.), the tiles below take
  // their rank-16 update.  1 100 MFMAs in all; what it costs is the 11 diagonal factorisations in sequence -- so wavefront 0
  // looks ahead: it takes the panel tile (j, j + 1), and while the others update the trailing tiles it updates (j + 1, j + 1)
  // alone and factorises it.
  {
    static_assert(DPB >= 2, "one wavefront factorises, the others update");
    double* scratch = lds + DPB * (DPB + 1) / 2 * TSZ;  // 16 x 16 block + 16 reciprocals for diag_block_factor
    const int nw = (int)blockDim.x >> 6;
    auto diag = [&](int j) {
      double* tjj = lds + tix(j, j) * TSZ;
      for (int e = l; e < 256; e += 64) scratch[e] = tjj[(e >> 4) * LDT + (e & 15)];
      tv::lds_sync();
      mf::diag_block_factor(scratch, nullptr, tjj, LDT);
      tv::lds_sync();
    };
    auto panel_tile = [&](int j, int jp) {  // U[j, jp] = W_j B[j, jp]
      const double* tjj = lds + tix(j, j) * TSZ;
      double* t = lds + tix(j, jp) * TSZ;
      mf::d4 r, u = mf::zero4();
#pragma unroll
      for (int v = 0; v < 4; ++v) r[v] = t[(4 * v + (l >> 4)) * LDT + (l & 15)];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) u = mf::mfma(tjj[(l & 15) * LDT + 4 * kk + (l >> 4)], r[kk], u);
#pragma unroll
      for (int v = 0; v < 4; ++v) t[(4 * v + (l >> 4)) * LDT + (l & 15)] = u[v];
    };
    auto trail_tile = [&](int j, int a, int b) {  // B[a, b] -= U[j, a]' U[j, b]
      const double* ua = lds + tix(j, a) * TSZ;
      const double* ub = lds + tix(j, b) * TSZ;
      double* tab = lds + tix(a, b) * TSZ;
      mf::d4 t;
#pragma unroll
      for (int v = 0; v < 4; ++v) t[v] = tab[(4 * v + (l >> 4)) * LDT + (l & 15)];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int o = (4 * ks + (l >> 4)) * LDT + (l & 15);
        t = mf::mfma(-ua[o], ub[o], t);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) tab[(4 * v + (l >> 4)) * LDT + (l & 15)] = t[v];
    };
    if (wave == 0) diag(0);
    __syncthreads();
    for (int j = 0; j + 1 < DPB; ++j) {
      if (wave == 0) panel_tile(j, j + 1);
      else
        for (int jp = j + 2 + (wave - 1); jp < DPB; jp += nw - 1) panel_tile(j, jp);
      __syncthreads();
      if (wave == 0) {
        trail_tile(j, j + 1, j + 1);
        tv::lds_sync();
        diag(j + 1);
      } else {
        const int m = DPB - 1 - j;
        for (int t = wave; t < m * (m + 1) / 2; t += nw - 1) {  // (tile 0 of the trailing block, (j + 1, j + 1), is wavefront 0's)
          int a = j + 1, rest = t;
          while (rest >= DPB - a) {
            rest -= DPB - a;
            ++a;
          }
          trail_tile(j, a, a + rest);
        }
      }
      __syncthreads();
    }
  }
  ODEF_SSTAMP(1);  // factorisation
  if constexpr (d % 4 == 0) {
    if (yfromx) {
      // X = P Sigma P, then Y'[:, c] = (At (x) I_d) X[:, c] in place: the rows a combination needs lie d apart, and d is a whole
      // number of the 4-row groups a register of the accumulator layout holds -- row group G = 4 t + v (rows 4 G .. 4 G + 3, one
      // derivative block J = 4 G / d) takes the groups G + (d / 4)(j - J), j > J, of the SAME lane.  Ascending G: sources lie ahead.
      // Same terms in the same order as smooth_predict_record / mfma_predict_phase.
      constexpr int GD = d / 4, NG = W::D / 4;
#pragma unroll
      for (int t = 0; t < DPB; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = t * mf::kB + 4 * v + (l >> 4);
          acc[t][v] *= pj_early[r] * pj_early[c0 + (l & 15)];
        }
      static_for<0, NG>([&](auto gc) {
        constexpr int G = decltype(gc)::value, J = (4 * G) / d;
        double y = acc[G / 4][G % 4];
        static_for<J + 1, q + 1>([&](auto jc) {
          constexpr int j = decltype(jc)::value, Gs = G + GD * (j - J);
          y += P.pc.At[J][j] * acc[Gs / 4][Gs % 4];
        });
        acc[G / 4][G % 4] = y;
      });
    }
  }
  ODEF_SSTAMP(2);  // (the right-hand sides are in the accumulators already)
  static_for<0, DPB>([&](auto jc) {  // forward: Z_j = W_j acc_j, acc_j' -= U[j, j']' Z_j for j' > j
    constexpr int j = decltype(jc)::value;
    const double* w = lds + tix(j, j) * TSZ;
    mf::d4 z0 = mf::zero4();
#pragma unroll
    for (int kk = 0; kk < (j == DPB - 1 ? KL : 4); ++kk) z0 = mf::mfma(w[(l & 15) * LDT + 4 * kk + (l >> 4)], acc[j][kk], z0);
    acc[j] = z0;
    const mf::d4 z = -z0;
    static_for<j + 1, DPB>([&](auto jpc) {
      constexpr int jp = decltype(jpc)::value;
      const double* t = lds + tix(j, jp) * TSZ;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc[jp] = mf::mfma(t[(4 * ks + (l >> 4)) * LDT + (l & 15)], z[ks], acc[jp]);
      asm volatile("" ::: "memory");  // keeps the compiler from hoisting (and spilling) the fragment reads of all later tiles
    });
  });
  ODEF_SSTAMP(3);  // forward sweep
  static_for<0, DPB>([&](auto jc) {  // backward: Gt_j = W_j' acc_j, acc_j' -= U[j', j] Gt_j for j' < j
    constexpr int j = DPB - 1 - decltype(jc)::value;
    const double* w = lds + tix(j, j) * TSZ;
    mf::d4 g0 = mf::zero4();
#pragma unroll
    for (int kk = 0; kk < (j == DPB - 1 ? KL : 4); ++kk) g0 = mf::mfma(w[(4 * kk + (l >> 4)) * LDT + (l & 15)], acc[j][kk], g0);
    acc[j] = g0;
    const mf::d4 g = -g0;
    static_for<0, j>([&](auto jpc) {
      constexpr int jp = decltype(jpc)::value;
      const double* t = lds + tix(jp, j) * TSZ;
#pragma unroll
      for (int ks = 0; ks < (j == DPB - 1 ? KL : 4); ++ks) acc[jp] = mf::mfma(t[(l & 15) * LDT + 4 * ks + (l >> 4)], g[ks], acc[jp]);
      asm volatile("" ::: "memory");
    });
  });
  ODEF_SSTAMP(4);  // backward sweep
  // What follows the sweeps, still on chip (smooth_onchip.h): G' never leaves the accumulators.
  //   m^s = P^-1 (P m + G delta)          (src/smoothing.jl:44, :26) -- the record and the carried mean of the pass
  //   Sigma^s = P^-1 (X + G M G') P^-1    M into the LDS the factor has left; the result tiles go straight to the record in
  //                                       the stage (packed lower triangle) and to the carried matrix SG (upper tiles, tile-major)
  using Pr = oc::Products<DPB>;
  constexpr int D = W::D;
  const size_t N = (size_t)P.N;
  const long s = P.split_sa;
  __syncthreads();  // every wavefront is done with the factor
  double* dl = lds + Pr::size;
  double* pij = dl + DP;
  double* pj = pij + DP;
  for (int k = tid; k < DP; k += (int)blockDim.x) {  // (zero behind the state dimension, as the predict kernel left them)
    dl[k] = my[W::DLV + k];
    pij[k] = my[W::PIJV + k];
    pj[k] = my[W::PJV + k];
  }
  __syncthreads();  // (the vectors are there)
  // M = P Sigma^s_+ P - B into the LDS the factor has left (swizzled upper tiles, smooth_onchip.h): Sigma^s_+ as this kernel
  // wrote it one record earlier (or the set-up did), B read a second time -- half of a wavefront's tiles at a time, G' keeps
  // the other registers
  {
    constexpr int NTU = Pr::NTU, PER = (NTU + DPB - 1) / DPB, HALF = (PER + 1) / 2;
    const double* SGr = my + W::SG;
#pragma unroll
    for (int h0 = 0; h0 < PER; h0 += HALF) {
      mf::d4 sg[HALF], bt[HALF];
#pragma unroll
      for (int u = h0; u < h0 + HALF && u < PER; ++u) {
        const int t = wave + u * DPB;
        int j = 0, rest = t;
        while (rest >= DPB - j) {
          rest -= DPB - j;
          ++j;
        }
        if (t < NTU) {
          sg[u - h0] = oc::load_tile_major(SGr + W::tile_at(j, j + rest));
          bt[u - h0] = oc::load_tile_major(BM + W::tile_at(j, j + rest));
        }
      }
#pragma unroll
      for (int u = h0; u < h0 + HALF && u < PER; ++u) {
        const int t = wave + u * DPB;
        int j = 0, rest = t;
        while (rest >= DPB - j) {
          rest -= DPB - j;
          ++j;
        }
        if (t < NTU) {
          double* dstm = lds + Pr::kM + t * 256;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int rr = 4 * v + (l >> 4), cc = l & 15;
            dstm[Pr::sw(rr, cc)] = sg[u - h0][v] * (pj[j * mf::kB + rr] * pj[(j + rest) * mf::kB + cc]) - bt[u - h0][v];
          }
        }
      }
    }
  }
  __syncthreads();
  ODEF_SSTAMP(5);  // M, vectors -> LDS
  {
    const double t = oc::gt_times<DPB>(acc, dl);
    const int k = c0 + (l & 15);
    if (l < 16 && k < D) {
      const double v = (my[W::MFV + k] + t) * pij[k];
      my[W::MSV + k] = v;
      P.smean[((size_t)s * D + k) * N + (size_t)i] = v;
      if (!(v == v)) P.retcode[i] = 3;  // "NaNs after smoothing" (src/smoothing.jl:25)
    }
  }
  ODEF_SSTAMP(6);  // mean
  mf::d4 r[Pr::WMAX];
  oc::gmgt<DPB, KL>(acc, lds, r);
  ODEF_SSTAMP(7);  // G M G'
  // X = P Sigma_s P comes from the record itself (packed lower triangle, still the filter's): the tile below the diagonal of
  // each pair, whole rows of it contiguous
  double* SG = my + W::SG;
  double* dst = P.stage + ((size_t)(s - P.stage_s0) * N + (size_t)i) * (size_t)P.stage_ld;
  mf::d4 x[Pr::WMAX];
#pragma unroll
  for (int w = 0; w < Pr::WMAX; ++w) {
    const int cw = wave + w < DPB ? wave + w : wave + w - DPB;
    if (w < Pr::owned(wave)) {
      const int tr = cw > wave ? cw : wave, tc = cw > wave ? wave : cw;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int a = tr * mf::kB + 4 * v + (l >> 4), b = tc * mf::kB + (l & 15);
        x[w][v] = (a < D && b <= a) ? dst[a * (a + 1) / 2 + b] * (pj[a] * pj[b]) : 0.0;
      }
    }
  }
  __syncthreads();  // every wavefront is done with M and the row buffer: each takes 16 x 17 doubles of LDS to transpose its tiles in
  double* tr = lds + wave * TSZ;
  auto transposed = [&](const mf::d4& t) {
    mf::d4 o;
#pragma unroll
    for (int v = 0; v < 4; ++v) tr[(4 * v + (l >> 4)) * LDT + (l & 15)] = t[v];
    tv::lds_sync();
#pragma unroll
    for (int v = 0; v < 4; ++v) o[v] = tr[(l & 15) * LDT + 4 * v + (l >> 4)];
    tv::lds_sync();
    return o;
  };
#pragma unroll
  for (int w = 0; w < Pr::WMAX; ++w) {
    if (w < Pr::owned(wave)) {
      // tile (cw, wave) of the sum and, through LDS, its transpose (wave, cw): whole rows leave.  The record
      // takes whichever of the two lies below the diagonal -- the transpose if the window wrapped (then X was read as that
      // transpose too).
      const int cw = wave + w < DPB ? wave + w : wave + w - DPB;
      const mf::d4 xw = cw >= wave ? x[w] : transposed(x[w]);
      mf::d4 o;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int a = cw * mf::kB + 4 * v + (l >> 4), b = c0 + (l & 15);
        o[v] = (xw[v] + r[w][v]) * (pij[a] * pij[b]);
      }
      const mf::d4 ot = transposed(o);
      if (cw == wave) {  // a diagonal tile: its lower triangle is what both halves get (the record holds nothing else)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int a = c0 + 4 * v + (l >> 4), b = c0 + (l & 15);
          if (b > a) o[v] = ot[v];
          if (b <= a && a < D) dst[a * (a + 1) / 2 + b] = o[v];
        }
        oc::store_tile_major(SG + W::tile_at(wave, wave), o);
      } else {
        const bool lower = cw > wave;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int a = (lower ? cw * mf::kB : c0) + 4 * v + (l >> 4), b = (lower ? c0 : cw * mf::kB) + (l & 15);
          if (a < D) dst[a * (a + 1) / 2 + b] = lower ? o[v] : ot[v];
        }
        oc::store_tile_major(SG + (lower ? W::tile_at(wave, cw) : W::tile_at(cw, wave)), lower ? ot : o);  // (the tile above the diagonal)
      }
    }
  }
  ODEF_SSTAMP(8);  // X + R, record, carried matrix
}
inline bool pleiades_smooth_split() {  // the staged pass as a sequence of kernels per record (default); ODEF_SMOOTH_SPLIT=0: one persistent launch per block
  const char* e = getenv("ODEF_SMOOTH_SPLIT");
  return !(e && e[0] == '0');
}
#ifndef ODEF_HOST_EMUL
// Dense output for the workgroup-per-trajectory path (dense_mfma.h): items = (trajectory, query time), grid-strided over
// gridDim.x workspaces of the MFMA smoother's size
template <int d, int q>
__global__ __launch_bounds__(kTeamBig, 2) void dense_mfma_kernel(const DenseParams P, double* ws) {
  using W = MfmaSmoothWs<d, q + 1>;
  __shared__ double lds[W::lds_size];
  double* my = ws + (size_t)blockIdx.x * W::size;
  for (size_t e = threadIdx.x; e < W::size; e += blockDim.x) my[e] = 0.0;  // padding rows / columns stay zero from here on
  __syncthreads();
  const long items = P.N * P.n_q;
  for (long it = (long)blockIdx.x; it < items; it += (long)gridDim.x) {
    dense_mfma_item<d, q>(P, it % P.N, it / P.N, my, lds);
    __syncthreads();
  }
}
// Posterior sampling for the workgroup-per-trajectory path (sample_mfma.h): items = (trajectory, sample), same grid stride
template <int d, int q>
__global__ __launch_bounds__(kTeamBig, 2) void sample_mfma_kernel(const SampleParams P, double* ws) {
  using W = MfmaSmoothWs<d, q + 1>;
  __shared__ double lds[W::lds_size];
  double* my = ws + (size_t)blockIdx.x * W::size;
  for (size_t e = threadIdx.x; e < W::size; e += blockDim.x) my[e] = 0.0;  // padding and the zero "next" covariance (SG)
  __syncthreads();
  const long items = P.N * P.n_samples;
  for (long it = (long)blockIdx.x; it < items; it += (long)gridDim.x) {
    sample_mfma_item<d, q>(P, it % P.N, it / P.N, my, lds);
    __syncthreads();
  }
}
constexpr long kDenseMfmaMaxGrid = 1024;
struct LaunchTeamSample {
  const SampleParams& P;
  double* ws;
  hipStream_t s;
  template <int d, int q>
  void operator()() {
    const long items = P.N * P.n_samples;
    const unsigned grid = (unsigned)(items < kDenseMfmaMaxGrid ? items : kDenseMfmaMaxGrid);
    hipLaunchKernelGGL((sample_mfma_kernel<d, q>), dim3(grid), dim3(kTeamBig), 0, s, P, ws);
  }
};
struct LaunchTeamDense {
  const DenseParams& P;
  double* ws;
  hipStream_t s;
  template <int d, int q>
  void operator()() {
    const long items = P.N * P.n_q;
    const unsigned grid = (unsigned)(items < kDenseMfmaMaxGrid ? items : kDenseMfmaMaxGrid);
    hipLaunchKernelGGL((dense_mfma_kernel<d, q>), dim3(grid), dim3(kTeamBig), 0, s, P, ws);
  }
};
#endif

struct LaunchTeamSmoothPredict {
  const SmoothParams& P;
  double* ws;
  hipStream_t s;
  int rc = 0;
  template <int d, int q>
  void operator()() {
    using W = MfmaSmoothWs<d, q + 1>;
    constexpr size_t lds_bytes = ((size_t)W::D * (W::D + 1) / 2 + 2 * W::DP) * sizeof(double);  // the packed record, P m, m^s_+
    static_assert(lds_bytes <= 160 * 1024, "the packed record does not fit the LDS");
    if (hipFuncSetAttribute((const void*)rts_smooth_predict_kernel<d, q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
      rc = -6;
      return;
    }
    hipLaunchKernelGGL((rts_smooth_predict_kernel<d, q>), dim3(team_grid(P.N)), dim3(predict_block<d>()), lds_bytes, s, P, ws);
  }
};
struct LaunchTeamSmoothSweeps {
  const SmoothParams& P;
  double* ws;
  hipStream_t s;
  int rc = 0;
  template <int d, int q>
  void operator()() {
    using W = MfmaSmoothWs<d, q + 1>;
    // the factor (tile rows padded to 17 doubles) and its scratch; then M, the row buffer(s) of Z and delta (smooth_onchip.h)
    constexpr size_t lds_factor = (size_t)(W::DPB * (W::DPB + 1) / 2) * mf::kB * 17 + 272, lds_products = (size_t)oc::Products<W::DPB>::size + 3 * W::DP;
    constexpr size_t lds_bytes = (lds_factor > lds_products ? lds_factor : lds_products) * sizeof(double);
    static_assert(lds_bytes <= 160 * 1024, "the on-chip record step does not fit the LDS");
    // (set at every launch: the attribute belongs to the current device, and a group of contexts spans several)
    if (hipFuncSetAttribute((const void*)rts_smooth_sweeps_kernel<d, q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
      rc = -6;
      return;
    }
    note_kernel("odef::rts_smooth_sweeps_kernel<%d, %d>", d, q);
    hipLaunchKernelGGL((rts_smooth_sweeps_kernel<d, q>), dim3(team_grid(P.N)), dim3(64 * W::DPB), lds_bytes, s, P, ws);
  }
};
struct LaunchTeamSmooth {
  const SmoothParams& P;
  double* ws;
  hipStream_t s;
  template <int d, int q>
  void operator()() {
    // (the split pass names its dominant kernel, rts_smooth_sweeps_kernel; this one only sets up its blocks)
    if (P.split_mode == 0) note_kernel("odef::rts_smooth_mfma_kernel<%d, %d, false>", d, q);
    if (P.split_mode != 0)
      hipLaunchKernelGGL((rts_smooth_mfma_kernel<d, q, true>), dim3(team_grid(P.N)), dim3(kTeamBig), 0, s, P, ws);
    else
      hipLaunchKernelGGL((rts_smooth_mfma_kernel<d, q, false>), dim3(team_grid(P.N)), dim3(kTeamBig), 0, s, P, ws);
  }
};

struct LaunchFilter {
  const FilterParams& P;
  int adaptive;
  hipStream_t s;
  template <class RHS, int q, bool EK1>
  void operator()() {
    const unsigned grid = (unsigned)((P.N + kWave - 1) / kWave);
    if constexpr (RHS::d * (q + 1) <= kRowsMaxD) {
      if (P.N < filter_rows_max_n()) {  // small ensemble: 16 lanes per trajectory
        const unsigned rgrid = rows_grid(P.N);
        if (adaptive) {
          note_kernel("odef::ek_filter_rows_adaptive_kernel<odef::%s, %d, %s>", RHS::name, q, tf(EK1));
          hipLaunchKernelGGL((ek_filter_rows_adaptive_kernel<RHS, q, EK1>), dim3(rgrid), dim3(kRowsBlock), 0, s, P);
        } else {
          note_kernel("odef::ek_filter_rows_kernel<odef::%s, %d, %s, %s>", RHS::name, q, tf(EK1), tf(P.everystep));
          if (P.everystep) hipLaunchKernelGGL((ek_filter_rows_kernel<RHS, q, EK1, true>), dim3(rgrid), dim3(kRowsBlock), 0, s, P);
          else hipLaunchKernelGGL((ek_filter_rows_kernel<RHS, q, EK1, false>), dim3(rgrid), dim3(kRowsBlock), 0, s, P);
        }
        return;
      }
    }
    if (adaptive) {
      note_kernel("odef::ek_filter_adaptive_kernel<odef::%s, %d, %s>", RHS::name, q, tf(EK1));
      hipLaunchKernelGGL((ek_filter_adaptive_kernel<RHS, q, EK1>), dim3(grid), dim3(kWave), 0, s, P);
      return;
    }
    const bool lag = P.everystep && P.N < filter_lag_max_n();  // small ensemble: spread the record stores over the next step
    note_kernel("odef::ek_filter_fixed_kernel<odef::%s, %d, %s, %s, %s>", RHS::name, q, tf(EK1), tf(P.everystep), tf(lag));
    if (lag) hipLaunchKernelGGL((ek_filter_fixed_kernel<RHS, q, EK1, true, true>), dim3(grid), dim3(kWave), 0, s, P);
    else if (P.everystep) hipLaunchKernelGGL((ek_filter_fixed_kernel<RHS, q, EK1, true>), dim3(grid), dim3(kWave), 0, s, P);
    else hipLaunchKernelGGL((ek_filter_fixed_kernel<RHS, q, EK1, false>), dim3(grid), dim3(kWave), 0, s, P);
  }
};
struct LaunchSmooth {
  const SmoothParams& P;
  hipStream_t s;
  template <int d, int q>
  void operator()() {
    constexpr int TPB = kWave / SmoothTeam<d * (q + 1)>::lanes;
    // Small state AND a large ensemble: one lane per trajectory.  A small ensemble does not fill the chip that
    // way (N / 64 wavefronts for 1 024 SIMDs); the row-per-lane team kernel gives TPB x fewer trajectories per
    // wavefront, i.e. more wavefronts, and wins below kSmoothLaneMinN.
    if constexpr (d * (q + 1) <= kRowsMaxD) {
      if (P.N < smooth_rows_max_n()) {
        const unsigned rgrid = rows_grid(P.N);
        note_kernel("odef::rts_smooth_bcast_kernel<%d, %d, %s>", d, q, tf(P.adaptive));
        if (P.adaptive)
          hipLaunchKernelGGL((rts_smooth_bcast_kernel<d, q, true>), dim3(rgrid), dim3(kRowsBlock), 0, s, P);
        else
          hipLaunchKernelGGL((rts_smooth_bcast_kernel<d, q, false>), dim3(rgrid), dim3(kRowsBlock), 0, s, P);
        return;
      }
    }
    bool lane_kernel = false;
    if constexpr (d * (q + 1) <= kSmoothLaneMaxD) lane_kernel = P.N >= smooth_lane_min_n();
    if constexpr (d * (q + 1) <= kSmoothLaneMaxD) {
      if (lane_kernel) {
        const unsigned grid = (unsigned)((P.N + kWave - 1) / kWave);
        note_kernel("odef::rts_smooth_lane_kernel<%d, %d, %s>", d, q, tf(P.adaptive));
        if (P.adaptive)
          hipLaunchKernelGGL((rts_smooth_lane_kernel<d, q, true>), dim3(grid), dim3(kWave), 0, s, P);
        else
          hipLaunchKernelGGL((rts_smooth_lane_kernel<d, q, false>), dim3(grid), dim3(kWave), 0, s, P);
        return;
      }
    }
    const unsigned grid = (unsigned)((P.N + TPB - 1) / TPB);
    note_kernel("odef::rts_smooth_kernel<%d, %d>", d, q);
    hipLaunchKernelGGL((rts_smooth_kernel<d, q>), dim3(grid), dim3(kWave), 0, s, P);
  }
};

}  // namespace odef
