// Cross-lane primitives for the 16-lanes-per-trajectory kernels (csrc/team_vec.h), measured on gfx950 with ONE wave per
// SIMD (the situation of a 4 096-trajectory ensemble): cycles per instruction (s_memtime ticks are 100 MHz, so the
// program reports wall time per instruction and the cycle count at the clock the chip actually ran) for
//   fma        v_fma_f64, 16 independent chains
//   chain      v_fma_f64, ONE dependent chain (latency)
//   bcast+fma  v_mov_b64_dpp row_newbcast + v_fma_f64 (what the compiler emits for the builtin)
//   fmac_dpp   v_fmac_f64_dpp row_newbcast (inline asm, one instruction)
//   shl32      2 x v_mov_b32_dpp row_shl:3 (a 64-bit row shift) + v_fma_f64
//   lds        ds_write_b64 / ds_read_b64 round trip (dependent)
// and checks that v_fmac_f64_dpp computes what the two-instruction form computes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/dpp_bench.hip -o tools/dpp_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

template <int K>
__device__ inline double bc(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + K, 0xf, 0xf, false);
}
template <int K>
__device__ inline void fmac_bc(double& acc, double src, double b) {  // acc += bcast_K(src) * b
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(b), "n"(K));
}
template <int K>
__device__ inline void fmac_bc_nonop(double& acc, double src, double b) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(b), "n"(K));
}
__device__ inline double shl3(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x103, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x103, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void kern(double* out, int iters) {
  __shared__ double lds[64 * 4 * 17];
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = threadIdx.x * 1e-3 + j;
  double b = 1.0000001 + threadIdx.x * 1e-9, c = 1e-9;
  double* my = lds + threadIdx.x * 17;
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_fma(a[j], b, c);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[0] = __builtin_fma(a[0], b, c);
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_fma(bc<3>(a[(j + 5) & 15]), c, a[j]);
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 16; ++j) fmac_bc<3>(a[j], a[(j + 5) & 15], c);
    } else if constexpr (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_fma(shl3(a[(j + 5) & 15]), c, a[j]);
    } else if constexpr (MODE == 5) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        my[j] = a[0];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        a[0] = lds[(threadIdx.x ^ 1) * 17 + j] + c;
      }
    } else if constexpr (MODE == 6) {
#pragma unroll
      for (int j = 0; j < 16; ++j) fmac_bc_nonop<3>(a[j], a[(j + 5) & 15], c);
    } else if constexpr (MODE == 7) {  // dependent chain THROUGH the dpp source: write then immediately read via DPP
#pragma unroll
      for (int j = 0; j < 16; ++j) a[0] = __builtin_fma(bc<3>(a[0]), c, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += a[j];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// correctness of the one-instruction form against the two-instruction form
__global__ void check(double* out) {
  const double x = 1.0 + threadIdx.x * 0.25, y = 3.0 - threadIdx.x * 0.125;
  double r1 = 0.5, r2 = 0.5;
  r1 = __builtin_fma(bc<5>(x), y, r1);
  r1 = __builtin_fma(-bc<11>(x), y, r1);
  fmac_bc<5>(r2, x, y);
  double t = -x;
  fmac_bc<11>(r2, t, y);
  out[threadIdx.x] = r1;
  out[64 + threadIdx.x] = r2;
}

int main() {
  double* d;
  hipMalloc(&d, (size_t)1024 * 256 * 8);
  double h[128];
  check<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int row = l & ~15;
    const double x5 = 1.0 + (row + 5) * 0.25, x11 = 1.0 + (row + 11) * 0.25, y = 3.0 - l * 0.125;
    const double ref = std::fma(-x11, y, std::fma(x5, y, 0.5));
    if (h[l] != ref || h[64 + l] != ref) ++bad;
  }
  printf("v_fmac_f64_dpp row_newbcast vs reference: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[] = {"fma x16 indep", "fma chain", "mov_dpp+fma", "s_nop1+fmac_dpp", "2x shl32+fma", "lds roundtrip", "fmac_dpp no nop", "dpp chain"};
  const double instr[] = {16, 16, 32, 16, 48, 16, 16, 32};
  for (int wps : {1, 2}) {
    for (int mode = 0; mode < 8; ++mode) {
      auto launch = [&](int it) {
        const int threads = 64 * 4 * wps;
        switch (mode) {
          case 0: kern<0><<<256, threads>>>(d, it); break;
          case 1: kern<1><<<256, threads>>>(d, it); break;
          case 2: kern<2><<<256, threads>>>(d, it); break;
          case 3: kern<3><<<256, threads>>>(d, it); break;
          case 4: kern<4><<<256, threads>>>(d, it); break;
          case 5: kern<5><<<256, threads>>>(d, it); break;
          case 6: kern<6><<<256, threads>>>(d, it); break;
          case 7: kern<7><<<256, threads>>>(d, it); break;
        }
      };
      launch(100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch(iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double ns_per_group = ms * 1e6 / (iters * 16.0);  // per loop-body element (one "operation" of the mode)
      printf("%-16s %d wave(s)/SIMD: %8.3f ms  %7.2f ns per op per wave  (~%.1f cycles @2.0 GHz, %.0f instr per op)\n",
             names[mode], wps, ms, ns_per_group, ns_per_group * 2.0, instr[mode] / 16.0);
    }
  }
  return 0;
}
