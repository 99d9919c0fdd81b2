// Internal interface of the run-time compilation layer (jit.hip) used by the C-ABI layer (api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace odef {

constexpr int kJitFirstId = 100;  // rhs ids >= 100 are run-time compiled vector fields

struct JitModule {
  hipModule_t mod = nullptr;
  hipFunction_t fixed_every = nullptr, fixed_final = nullptr, adaptive = nullptr;
  hipFunction_t smooth_fixed = nullptr, smooth_adapt = nullptr, dense = nullptr, sample = nullptr;
  hipFunction_t smooth_rows = nullptr;  // 12 < state dimension <= 32: row-per-lane team smoother
  hipFunction_t dense_rows = nullptr;   // ... and dense output on the same teams (one team per (trajectory, query time))
  hipFunction_t sample_rows = nullptr;  // ... and posterior sampling (one team per (trajectory, sample))
  int rows_team = 16;                   // lanes per trajectory of that kernel
  // state dimension <= 16: the 16-lanes-per-trajectory kernels of small and sharded ensembles (rows_kernels.h), workgroups of 256
  hipFunction_t rows_fixed_every = nullptr, rows_fixed_final = nullptr, rows_adaptive = nullptr;
  hipFunction_t bcast_fixed = nullptr, bcast_adapt = nullptr;
  bool rows16 = false;
  bool posterior = false;  // lane smoother / dense output / sampler available (state dimension <= 12)
};

// returns the new rhs id (>= kJitFirstId) or -1 with the compiler log in `err`
int jit_register(const char* name, const char* source, int d, int np, const char* include_dir, std::string& err);
bool jit_lookup(int rhs_id, int* d, int* np);
// compiles (once per (rhs, order, alg, device)) and loads the kernels on the CURRENT device
JitModule* jit_get_module(int rhs_id, int q, int ek1, int device, std::string& err);
// The workgroup-per-trajectory path for a run-time compiled field (state dimension above 20, even d <= 32): compiles
// (once per (rhs, order, alg); minutes) a host + device shared object around the field and returns its launch table (launch.h)
struct TeamLaunch;
const TeamLaunch* jit_get_team(int rhs_id, int q, int ek1, unsigned long abi_stamp, std::string& err);  // abi_stamp: team_abi_stamp() of the library
// `block` threads per workgroup (64: the lane and LDS row-team kernels; 256: rows_kernels.h); params: pointer to the kernel's
// single by-value parameter struct
int jit_launch(hipFunction_t f, unsigned gx, unsigned gy, const void* params, hipStream_t s, unsigned block = 64);

}  // namespace odef
