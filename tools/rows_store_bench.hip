// What does the record store of the 16-lanes-per-trajectory kernels cost, and which store shape is cheapest?
// 1 024 wavefronts (one per SIMD, the 4 096-trajectory case), each step = `work` dependent-free FP64 FMAs followed by
// the stores of one record (D = 12: 12 mean + 78 covariance + 1 diffusion doubles per trajectory, layout
// [slot][element][N]) in one of these shapes:
//   0 none      compute only
//   1 tri       what a row-per-lane kernel does naturally: lane r of team t stores X[r][c], c = 0..11 (only c <= r in
//               range): 14 instructions, 91 segments of 32 B (4 trajectories x 8 B)
//   2 packed    the same 91 segments packed 16 to an instruction: 6 instructions
//   3 wg-lines  four wavefronts (16 trajectories) share the record: every instruction writes four full 128-B lines,
//               23 instructions per workgroup-step (~6 per wave-step)
//   4 tri-nt    shape 1 with non-temporal stores
// Build: hipcc --offload-arch=gfx950 -O3 tools/rows_store_bench.hip -o tools/rows_store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int D = 12, TRI = 78, REC = D + TRI + 1;

__device__ inline int tri(int i, int j) { return i * (i + 1) / 2 + j; }

template <int SHAPE>
__global__ __launch_bounds__(256) void kern(double* out, long N, int nsteps, int work) {
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64, team = lane / 16, r = lane % 16;
  const long wave_id = (long)blockIdx.x * (blockDim.x / 64) + wave;
  const long i = wave_id * 4 + team;  // trajectory of this team
  double a[8];
  for (int j = 0; j < 8; ++j) a[j] = 1.0 + lane * 1e-3 + j;
  const double b = 1.0000001, c = 1e-9;
  // per-lane element offsets (bytes within one slot), 0xFFFFFFFF = out of range (dropped by the buffer bounds check)
  unsigned off[16];
  for (int k = 0; k < 16; ++k) off[k] = 0xFFFFFFFFu;
  if (SHAPE == 1 || SHAPE == 4) {
    if (r < D) off[0] = (unsigned)(((size_t)r * N + i) * 8);  // mean
    for (int cc = 0; cc < D; ++cc)
      if (r < D && cc <= r) off[1 + cc] = (unsigned)(((size_t)(D + tri(r, cc)) * N + i) * 8);
    if (r == 0) off[13] = (unsigned)(((size_t)(D + TRI) * N + i) * 8);
  } else if (SHAPE == 2) {
    for (int k = 0; k < 6; ++k) {
      const int e = 16 * k + r;
      if (e < REC) off[k] = (unsigned)(((size_t)e * N + i) * 8);
    }
  } else if (SHAPE == 3) {
    // workgroup of 4 waves = 16 trajectories i16 .. i16+15: element e is one 128-B line; thread t of the workgroup
    // handles (line = 16 k' + t / 16 ..., traj = t % 16): 256 threads cover 16 lines per round, 6 rounds for 91 lines
    const long i16 = (long)blockIdx.x * 16;
    for (int k = 0; k < 6; ++k) {
      const int e = 16 * k + (int)threadIdx.x / 16;
      if (e < REC) off[k] = (unsigned)(((size_t)e * N + i16 + threadIdx.x % 16) * 8);
    }
  }
  for (int n = 0; n < nsteps; ++n) {
    for (int w = 0; w < work / 8; ++w) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], b, c);
    }
    if (SHAPE != 0) {
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)n * REC * N), 0, (int)((size_t)REC * N * 8), 0x00020000);
      constexpr int NI = (SHAPE == 1 || SHAPE == 4) ? 14 : 6;
#pragma unroll
      for (int k = 0; k < NI; ++k)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a[k % 8]), rs, off[k], 0, SHAPE == 4 ? 2 : 0);
    }
  }
  double s = 0;
  for (int j = 0; j < 8; ++j) s += a[j];
  if (s == 12345.678) out[0] = s;
}

template <int SHAPE>
float run(double* d, long N, int nsteps, int work, int wg_waves) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  const int waves = (int)(N / 4);
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((kern<SHAPE>), dim3(waves / wg_waves), dim3(64 * wg_waves), 0, 0, d, N, nsteps, work);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const int nsteps = 1024;
  for (long N : {4096L, 16384L}) {
    double* d; const size_t bytes = (size_t)(nsteps + 1) * REC * N * 8;
    CK(hipMalloc((void**)&d, bytes));
    for (int work : {0, 600}) {
      const float t0 = run<0>(d, N, nsteps, work, 1);
      const float t1 = run<1>(d, N, nsteps, work, 1);
      const float t2 = run<2>(d, N, nsteps, work, 1);
      const float t3 = run<3>(d, N, nsteps, work, 4);
      const float t4 = run<4>(d, N, nsteps, work, 1);
      const float t0w = run<0>(d, N, nsteps, work, 4);
      printf("N=%ld work=%d FMAs/step: none %.3f ms (4-wave WG %.3f) | tri(14 instr) %.3f | packed(6 instr) %.3f | wg-lines(6 instr, full lines) %.3f | tri-nt %.3f   [%.2f GB per run]\n",
             N, work, t0, t0w, t1, t2, t3, t4, bytes / 1e9);
    }
    CK(hipFree(d));
  }
  return 0;
}
