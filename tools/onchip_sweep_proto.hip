// Prototype for DESIGN 7.1: the two block sweeps  G' = (U'U)^-1 Y'  of the D = 168 smoother with the factor U in LDS and the
// right-hand sides in accumulator registers -- one workgroup of 11 wavefronts per CU, wavefront c owns tile column c of Y'
// (11 tiles = 88 registers), no barrier inside, A operands read from LDS (tile rows padded to 17 doubles: the transposed
// reads of the backward sweep are bank-conflict free).  Measures the time of a sweep pair per workgroup against the
// 140 us per step and CU the global-workspace sweeps of csrc/mfma_dense.h cost, and checks the result on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I odefilters.jl_amd/csrc tools/onchip_sweep_proto.hip -o tools/onchip_sweep_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "mfma_dense.h"
using namespace odef;
using mf::d4;
constexpr int DPB = 11, kB = 16, DP = DPB * kB, LDT = 17, TSZ = kB * LDT, NTU = DPB * (DPB + 1) / 2;
__host__ __device__ constexpr int tix(int j, int jp) { return j * DPB - j * (j - 1) / 2 + (jp - j); }  // upper tile (j, jp >= j)

template <bool TRANS>
__device__ __attribute__((always_inline)) inline double wf(const double* w, int kk) {
  const int l = threadIdx.x & 63;
  return TRANS ? w[(4 * kk + (l >> 4)) * LDT + (l & 15)] : w[(l & 15) * LDT + 4 * kk + (l >> 4)];
}
template <bool TRANS>
__device__ __attribute__((always_inline)) inline d4 applyw(const double* w, d4 r) {
  d4 o = mf::zero4();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) o = mf::mfma(wf<TRANS>(w, kk), r[kk], o);
  return o;
}

__global__ __launch_bounds__(64 * DPB) void k_sweeps(const double* __restrict__ Ut /* [NTU][16][16], diagonal tiles hold W_j */,
                                                     const double* __restrict__ Y, double* __restrict__ G, int iters) {
  extern __shared__ double lds[];  // NTU tiles of 16 x 17
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
  for (int e = tid; e < NTU * 256; e += blockDim.x) lds[(e >> 8) * TSZ + ((e >> 4) & 15) * LDT + (e & 15)] = Ut[e];
  __syncthreads();
  const double* Yw = Y + (size_t)blockIdx.x * DP * DP;
  double* Gw = G + (size_t)blockIdx.x * DP * DP;
  const int c0 = wave * kB;
  for (int it = 0; it < iters; ++it) {
    d4 acc[DPB];
#pragma unroll
    for (int j = 0; j < DPB; ++j) acc[j] = mf::load_tile(Yw, DP, j * kB, c0);
    static_for<0, DPB>([&](auto jc) {  // forward
      constexpr int j = decltype(jc)::value;
      const d4 z0 = applyw<false>(lds + tix(j, j) * TSZ, acc[j]);
      acc[j] = z0;
      const d4 z = -z0;
      static_for<j + 1, DPB>([&](auto jpc) {
        constexpr int jp = decltype(jpc)::value;
        const double* t = lds + tix(j, jp) * TSZ;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[jp] = mf::mfma(t[(4 * ks + (l >> 4)) * LDT + (l & 15)], z[ks], acc[jp]);
        if constexpr ((jp - j) % 1 == 0) asm volatile("" ::: "memory");  // keeps the compiler from hoisting (and spilling) all fragment reads
      });
    });
    static_for<0, DPB>([&](auto jc) {  // backward
      constexpr int j = DPB - 1 - decltype(jc)::value;
      const d4 g0 = applyw<true>(lds + tix(j, j) * TSZ, acc[j]);
      acc[j] = g0;
      const d4 g = -g0;
      static_for<0, j>([&](auto jpc) {
        constexpr int jp = decltype(jpc)::value;
        const double* t = lds + tix(jp, j) * TSZ;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[jp] = mf::mfma(t[(l & 15) * LDT + 4 * ks + (l >> 4)], g[ks], acc[jp]);
        if constexpr ((j - jp) % 1 == 0) asm volatile("" ::: "memory");
      });
    });
#pragma unroll
    for (int j = 0; j < DPB; ++j) mf::store_tile(Gw, DP, j * kB, c0, acc[j]);
  }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 20;
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  // SPD B = F F' + DP I, upper Cholesky U (B = U'U)
  std::vector<double> B(DP * DP, 0.0), U(DP * DP, 0.0), F(DP * 8);
  for (auto& x : F) x = nd(rng);
  for (int a = 0; a < DP; ++a)
    for (int b = 0; b < DP; ++b) {
      double s = a == b ? 4.0 : 0.0;
      for (int k = 0; k < 8; ++k) s += 0.1 * F[a * 8 + k] * F[b * 8 + k];
      B[a * DP + b] = s;
    }
  for (int i = 0; i < DP; ++i)
    for (int j = i; j < DP; ++j) {
      double s = B[i * DP + j];
      for (int k = 0; k < i; ++k) s -= U[k * DP + i] * U[k * DP + j];
      U[i * DP + j] = i == j ? std::sqrt(s) : s / U[i * DP + i];
    }
  // tiles: off-diagonal U tiles, diagonal tiles W_j = L_jj^-1 (L_jj = U_jj')
  std::vector<double> Ut((size_t)NTU * 256, 0.0);
  for (int j = 0; j < DPB; ++j)
    for (int jp = j; jp < DPB; ++jp)
      for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) Ut[(size_t)tix(j, jp) * 256 + r * 16 + c] = U[(j * 16 + r) * DP + jp * 16 + c];
  for (int j = 0; j < DPB; ++j) {
    double L[16][16], W[16][16] = {};
    for (int r = 0; r < 16; ++r)
      for (int c = 0; c < 16; ++c) L[r][c] = U[(j * 16 + c) * DP + j * 16 + r];
    for (int c = 0; c < 16; ++c)
      for (int r = c; r < 16; ++r) {
        double s = r == c ? 1.0 : 0.0;
        for (int k = c; k < r; ++k) s -= L[r][k] * W[k][c];
        W[r][c] = s / L[r][r];
      }
    for (int r = 0; r < 16; ++r)
      for (int c = 0; c < 16; ++c) Ut[(size_t)tix(j, j) * 256 + r * 16 + c] = W[r][c];
  }
  std::vector<double> Y((size_t)DP * DP), Gref((size_t)DP * DP);
  for (auto& x : Y) x = nd(rng);
  // reference: solve U'U G = Y column by column
  for (int c = 0; c < DP; ++c) {
    std::vector<double> z(DP), g(DP);
    for (int i = 0; i < DP; ++i) {
      double s = Y[i * DP + c];
      for (int k = 0; k < i; ++k) s -= U[k * DP + i] * z[k];
      z[i] = s / U[i * DP + i];
    }
    for (int i = DP - 1; i >= 0; --i) {
      double s = z[i];
      for (int k = i + 1; k < DP; ++k) s -= U[i * DP + k] * g[k];
      g[i] = s / U[i * DP + i];
    }
    for (int i = 0; i < DP; ++i) Gref[i * DP + c] = g[i];
  }
  double *dU, *dY, *dG;
  hipMalloc(&dU, Ut.size() * 8); hipMemcpy(dU, Ut.data(), Ut.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&dY, (size_t)nwg * DP * DP * 8); hipMalloc(&dG, (size_t)nwg * DP * DP * 8);
  for (int w = 0; w < nwg; ++w) hipMemcpy(dY + (size_t)w * DP * DP, Y.data(), Y.size() * 8, hipMemcpyHostToDevice);
  const size_t ldsb = (size_t)NTU * TSZ * 8;
  hipFuncSetAttribute((const void*)k_sweeps, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_sweeps<<<nwg, 64 * DPB, ldsb>>>(dU, dY, dG, 1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<double> Gh((size_t)DP * DP);
  hipMemcpy(Gh.data(), dG + (size_t)(nwg - 1) * DP * DP, Gh.size() * 8, hipMemcpyDeviceToHost);
  double err = 0, nrm = 0;
  for (size_t i = 0; i < Gh.size(); ++i) { err = std::fmax(err, std::fabs(Gh[i] - Gref[i])); nrm = std::fmax(nrm, std::fabs(Gref[i])); }
  hipEventRecord(e0);
  k_sweeps<<<nwg, 64 * DPB, ldsb>>>(dU, dY, dG, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int rounds = (nwg + 255) / 256;
  printf("LDS %zu KB; max |G - ref| / max |ref| = %.2e; %d workgroups x %d sweep pairs: %.3f ms = %.1f us per sweep pair per CU\n", ldsb >> 10,
         err / nrm, nwg, iters, ms, ms * 1e3 / (iters * rounds));
  return err / nrm < 1e-12 ? 0 : 1;
}
