// Prototype / micro-benchmark for csrc/smooth_onchip.h: R = G M G' with G' resident in the accumulators of DPB wavefronts,
// M in LDS and the rows of Z = M G' exchanged through an LDS row buffer.  Checks the result on the host and prints the time
// per workgroup (one workgroup per CU: the LDS holds one).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I odefilters.jl_amd/csrc tools/onchip_products_proto.hip -o /tmp/onchip_products_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "smooth_onchip.h"
using namespace odef;
using mf::d4;
#ifndef PROTO_DPB
#define PROTO_DPB 11
#endif
constexpr int DPB = PROTO_DPB, DP = DPB * 16;
using Pr = oc::Products<DPB>;

__global__ __launch_bounds__(64 * DPB) void k_products(const double* __restrict__ GT, const double* __restrict__ MM, double* __restrict__ R, int iters) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* g = GT + (size_t)blockIdx.x * DP * DP;
  const double* m = MM + (size_t)blockIdx.x * DP * DP;
  double* out = R + (size_t)blockIdx.x * DP * DP;
  for (int it = 0; it < iters; ++it) {
    d4 acc[DPB];
#pragma unroll
    for (int j = 0; j < DPB; ++j) acc[j] = mf::load_tile(g, DP, j * 16, wave * 16);
    __syncthreads();
    oc::load_m<DPB>(m, lds);
    __syncthreads();
    d4 r[Pr::WMAX];
    oc::gmgt<DPB>(acc, lds, r);
#pragma unroll
    for (int w = 0; w < Pr::WMAX; ++w) {
      if (w < Pr::owned(wave)) {
        const int cp = (wave + w) % DPB;
        mf::store_tile(out, DP, cp * 16, wave * 16, r[w]);
        if (cp != wave) mf::store_tile_t(out, DP, wave * 16, cp * 16, r[w]);
      }
    }
  }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 20;
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  std::vector<double> G((size_t)DP * DP), M((size_t)DP * DP), Z((size_t)DP * DP), Rref((size_t)DP * DP);
  for (auto& x : G) x = nd(rng);
  for (int a = 0; a < DP; ++a)
    for (int b = 0; b <= a; ++b) M[a * DP + b] = M[b * DP + a] = nd(rng);
  for (int a = 0; a < DP; ++a)
    for (int b = 0; b < DP; ++b) {
      double s = 0;
      for (int k = 0; k < DP; ++k) s += M[a * DP + k] * G[k * DP + b];
      Z[a * DP + b] = s;
    }
  for (int a = 0; a < DP; ++a)
    for (int b = 0; b < DP; ++b) {
      double s = 0;
      for (int k = 0; k < DP; ++k) s += Z[k * DP + a] * G[k * DP + b];
      Rref[a * DP + b] = s;
    }
  double *dG, *dM, *dR;
  hipMalloc(&dG, (size_t)nwg * DP * DP * 8); hipMalloc(&dM, (size_t)nwg * DP * DP * 8); hipMalloc(&dR, (size_t)nwg * DP * DP * 8);
  std::vector<double> Mt((size_t)DP * DP);  // tile-major, as the predict kernel leaves M (MfmaSmoothWs::tm)
  for (int a = 0; a < DP; ++a)
    for (int b = 0; b < DP; ++b) Mt[(size_t)((b / 16) * DPB + a / 16) * 256 + (a % 16) * 16 + b % 16] = M[a * DP + b];
  for (int w = 0; w < nwg; ++w) {
    hipMemcpy(dG + (size_t)w * DP * DP, G.data(), G.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dM + (size_t)w * DP * DP, Mt.data(), Mt.size() * 8, hipMemcpyHostToDevice);
  }
  hipMemset(dR, 0, (size_t)nwg * DP * DP * 8);
  const size_t ldsb = (size_t)Pr::size * 8;
  hipFuncSetAttribute((const void*)k_products, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_products<<<nwg, 64 * DPB, ldsb>>>(dG, dM, dR, 1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<double> Rh((size_t)DP * DP);
  hipMemcpy(Rh.data(), dR + (size_t)(nwg - 1) * DP * DP, Rh.size() * 8, hipMemcpyDeviceToHost);
  double err = 0, nrm = 0;
  for (size_t i = 0; i < Rh.size(); ++i) { err = std::fmax(err, std::fabs(Rh[i] - Rref[i])); nrm = std::fmax(nrm, std::fabs(Rref[i])); }
  hipEventRecord(e0);
  k_products<<<nwg, 64 * DPB, ldsb>>>(dG, dM, dR, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int rounds = (nwg + 255) / 256;
  const double mfmas = (double)DPB * (DPB * DPB + Pr::NTU) * 4;  // per workgroup
  printf("DPB %d, LDS %zu KB, %d row buffer(s); max |R - ref| / max |ref| = %.2e; %d workgroups x %d: %.3f ms = %.1f us per G M G' per CU (matrix-pipe floor %.1f us)\n",
         DPB, ldsb >> 10, Pr::NBUF, err / nrm, nwg, iters, ms, ms * 1e3 / (iters * rounds), mfmas * 64 / 4 / 2400.0);
  return err / nrm < 1e-12 ? 0 : 1;
}
