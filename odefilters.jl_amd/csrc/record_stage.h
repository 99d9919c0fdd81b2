// Covariance records between the solution layout and a trajectory-major stage.
//
// The solution arrays are [record][element][trajectory] (include/odefilter.h): what a lane-per-trajectory kernel reads
// and writes coalesced.  A workgroup-per-trajectory kernel (D = 168) touches ONE trajectory's 14 196 elements per record,
// i.e. 14 196 separate 128-byte lines for 8 useful bytes each; 1 024 resident workgroups at different steps share none of
// them in the 4 MB L2 of their XCD, so the 113 KB record costs 1.8 MB of HBM traffic (measured: a quarter of the smoother's
// time).  These two kernels move a block of records through LDS tiles to [record][trajectory][ld] and back, full lines on
// both sides, so the smoother reads and writes each record as one contiguous run.
#pragma once
#include <hip/hip_runtime.h>

namespace odef {

constexpr int kStageTile = 64;  // the kernels are templates on it only to be header-defined once per program

// grid (ceil(N / 64), ceil(n_el / 64), n_rec), 256 threads.  src [n_rec][n_el][N] -> dst [n_rec][N][ld]
template <int kTile>
__global__ __launch_bounds__(256) void stage_in_kernel(const double* __restrict__ src, double* __restrict__ dst, long N, long n_el,
                                                       long ld) {
  __shared__ double t[kTile][kTile + 1];
  const long i0 = (long)blockIdx.x * kTile, e0 = (long)blockIdx.y * kTile;
  const int tx = (int)threadIdx.x % kTile, ty = (int)threadIdx.x / kTile;
  src += (size_t)blockIdx.z * (size_t)n_el * (size_t)N;
  dst += (size_t)blockIdx.z * (size_t)N * (size_t)ld;
#pragma unroll 4
  for (int k = ty; k < kTile; k += 4) {
    const long e = e0 + k, i = i0 + tx;
    if (e < n_el && i < N) t[k][tx] = src[(size_t)e * N + i];
  }
  __syncthreads();
#pragma unroll 4
  for (int k = ty; k < kTile; k += 4) {
    const long i = i0 + k, e = e0 + tx;
    if (i < N && e < n_el) dst[(size_t)i * ld + e] = t[tx][k];
  }
}

// the way back: src [n_rec][N][ld] -> dst [n_rec][n_el][N]
template <int kTile>
__global__ __launch_bounds__(256) void stage_out_kernel(const double* __restrict__ src, double* __restrict__ dst, long N, long n_el,
                                                        long ld) {
  __shared__ double t[kTile][kTile + 1];
  const long i0 = (long)blockIdx.x * kTile, e0 = (long)blockIdx.y * kTile;
  const int tx = (int)threadIdx.x % kTile, ty = (int)threadIdx.x / kTile;
  src += (size_t)blockIdx.z * (size_t)N * (size_t)ld;
  dst += (size_t)blockIdx.z * (size_t)n_el * (size_t)N;
#pragma unroll 4
  for (int k = ty; k < kTile; k += 4) {
    const long i = i0 + k, e = e0 + tx;
    if (i < N && e < n_el) t[k][tx] = src[(size_t)i * ld + e];
  }
  __syncthreads();
#pragma unroll 4
  for (int k = ty; k < kTile; k += 4) {
    const long e = e0 + k, i = i0 + tx;
    if (e < n_el && i < N) dst[(size_t)e * N + i] = t[tx][k];
  }
}

inline long stage_record_ld(long n_el) { return (n_el + 15) / 16 * 16; }

}  // namespace odef
