// Microbenchmark: what HBM write rate does the filter kernel's store shape reach on its own?
// `waves` wavefronts, each storing field rows of 512 B (8 B/lane) or 1 KiB (16 B/lane) per step into a
// [nsteps][rows][N] array -- no arithmetic.  Variants: nontemporal stores, more waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int W, int NT>
__global__ __launch_bounds__(64) void k_store(double* out, long N, int rows, int nsteps, int split) {
  // `split` blocks share one 64-trajectory column group, each writing rows/split of the rows
  const long grp = blockIdx.x / split, part = blockIdx.x % split;
  const long i0 = grp * 64;
  const unsigned lane = threadIdx.x;
  double v = (double)lane;
  const int r0 = (int)part * (rows / split), r1 = r0 + rows / split;
  for (int n = 0; n < nsteps; ++n) {
    double* base = out + ((size_t)n * rows * N + (size_t)r0 * N + i0 * W);
    for (int k = r0; k < r1; k += W) {
      if (W == 1) { if (NT) __builtin_nontemporal_store(v, base + lane); else base[lane] = v; }
      else {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 t = {v, v + 1.0};
        if (NT) __builtin_nontemporal_store(t, (d2*)base + lane); else ((d2*)base)[lane] = t;
      }
      base += N * W;
    }
    v += 1.0;
  }
}

// Blocked layout [nsteps][N/64][rows][64]: the record of one wavefront (rows x 512 B) is one contiguous block.
template <int NT>
__global__ __launch_bounds__(64) void k_store_blocked(double* out, long N, int rows, int nsteps) {
  const long grp = blockIdx.x, G = N / 64;
  const unsigned lane = threadIdx.x;
  double v = (double)lane;
  for (int n = 0; n < nsteps; ++n) {
    double* base = out + (((size_t)n * G + grp) * rows) * 64;
    for (int k = 0; k < rows; ++k) {
      if (NT) __builtin_nontemporal_store(v, base + lane); else base[lane] = v;
      base += 64;
    }
    v += 1.0;
  }
}
template <int NT>
float run_blocked(double* d, long N, int rows, int nsteps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_store_blocked<NT>), dim3(N / 64), dim3(64), 0, 0, d, N, rows, nsteps);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

template <int W, int NT>
float run(double* d, long N, int rows, int nsteps, int split) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_store<W, NT>), dim3(N / 64 * split), dim3(64), 0, 0, d, N, rows, nsteps, split);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const long N = 65536; const int rows = 96, nsteps = argc > 1 ? atoi(argv[1]) : 512;
  double* d; size_t bytes = (size_t)nsteps * rows * N * 8;
  CK(hipMalloc((void**)&d, bytes));
  {
    float t;
    t = run_blocked<0>(d, N, rows, nsteps); printf("blocked layout, 1 wave/SIMD, 8 B/lane plain  %.3f ms %.2f TB/s\n", t, bytes / (t * 1e-3) / 1e12);
    t = run_blocked<1>(d, N, rows, nsteps); printf("blocked layout, 1 wave/SIMD, 8 B/lane nt     %.3f ms %.2f TB/s\n", t, bytes / (t * 1e-3) / 1e12);
  }
  for (int split : {1, 2, 4, 8}) {
    float t;
    t = run<1, 0>(d, N, rows, nsteps, split); printf("waves/SIMD=%d  8 B/lane plain  %.3f ms %.2f TB/s\n", split, t, bytes / (t * 1e-3) / 1e12);
    t = run<1, 1>(d, N, rows, nsteps, split); printf("waves/SIMD=%d  8 B/lane nt     %.3f ms %.2f TB/s\n", split, t, bytes / (t * 1e-3) / 1e12);
    t = run<2, 0>(d, N, rows, nsteps, split); printf("waves/SIMD=%d 16 B/lane plain  %.3f ms %.2f TB/s\n", split, t, bytes / (t * 1e-3) / 1e12);
    t = run<2, 1>(d, N, rows, nsteps, split); printf("waves/SIMD=%d 16 B/lane nt     %.3f ms %.2f TB/s\n", split, t, bytes / (t * 1e-3) / 1e12);
  }
  return 0;
}
