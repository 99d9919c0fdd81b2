// C-ABI layer of libodefilter_hip.so (see include/odefilter.h for the contract and the
// reference interfaces each entry point replaces).  Owns device buffers, builds the prior
// tables (src/priors.jl:7-59) and the per-step preconditioner seeds
// (src/preconditioning.jl:9) on the host, launches the gfx950 kernels, times them with
// hipEvents on the launch stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

#include "../../include/odefilter.h"
#include "jit.h"
#include "launch.h"

using namespace odef;

namespace {

thread_local std::string g_create_error;

struct Buf {
  void* ptr = nullptr;
  size_t bytes = 0;     // capacity
  size_t valid = 0;     // bytes holding results
  bool owned = false;
  bool bound = false;
};

struct RhsInfo { int d, np; };
const RhsInfo kRhs[] = {{2, 3}, {3, 3}, {2, 4}, {2, 1}, {2, 2}, {28, 0}, {16, 1}};

}  // namespace

struct odef_ctx {
  odef_config cfg{};
  int d = 0, q = 0, D = 0, TRI = 0, np = 0;
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  PriorConsts pc{};
  double* d_u0 = nullptr;
  double* d_p = nullptr;
  bool have_problem = false;
  double t0 = 0.0;
  // time grid of the last fixed solve
  std::vector<double> tgrid;
  double* d_hs = nullptr;
  double* d_tgrid = nullptr;
  bool grid_on_device = false;  // d_hs / d_ptab / d_tab_idx / d_tgrid hold the tables of `tgrid`
  double* d_tq = nullptr;
  size_t tq_cap = 0;
  long n_q = 0;
  long n_samples = 0;
  bool smoothed_done = false;
  double* d_ptab = nullptr;
  int* d_tab_idx = nullptr;
  size_t grid_cap = 0;
  long n_save = 0;
  bool adaptive = false;
  bool solved = false;
  bool team_path = false;   // workgroup-per-trajectory kernels (large state dimension)
  const TeamLaunch* team = nullptr;  // ... and their launch functions for this vector field
  JitModule* jit = nullptr; // run-time compiled vector field (rhs_id >= 100); owned by the registry in jit.hip
  double* d_ws = nullptr;   // per-trajectory workspace of the team kernels
  double* d_stage = nullptr;  // trajectory-major stage of the covariance records (D = 168 smoother, record_stage.h)
  long stage_filter_recs = 0; // > 0: the last solve left this many filter records in the stage (record r at r N ld)
  size_t stage_cap = 0;     // doubles
  size_t ws_cap = 0;
  Buf f[ODEF_F_COUNT_];
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float ms[2] = {0.f, 0.f};
  int nl[2] = {0, 0};
  char kname[2][192] = {"", ""};  // kernel of the last filter / smoother pass (odef_kernel_name)
  // odef_group runs its shards concurrently: with `defer` set, odef_solve_* / odef_smooth return after the launch and
  // complete_pending() does the wait + timing (pending: 1 = filter, 2 = smoother)
  bool defer = false;
  int pending = 0;
  std::string err;
};

namespace {

int fail(odef_ctx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  else g_create_error = buf;
  return -1;
}

#define HIPCHK(c, call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) return fail((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ibm(d,q) reduced to its (q+1)x(q+1) scalar blocks (src/priors.jl:15-56).
void build_prior(int q, PriorConsts& pc) {
  std::memset(&pc, 0, sizeof pc);
  const int nb = q + 1;
  for (int J = 0; J < nb; ++J) {
    pc.At[J][J] = 1.0;
  }
  {
    double val = 1.0;
    for (int i = 1; i <= q; ++i) {  // A[j, j+d*i] = 1/i!  by the reference's running division
      val = val / i;
      for (int J = 0; J + i < nb; ++J) pc.At[J][J + i] = val;
    }
  }
  auto fact = [](int n) { double f = 1.0; for (int k = 2; k <= n; ++k) f *= k; return f; };
  for (int c = 0; c < nb; ++c)
    for (int r = c; r < nb; ++r) {
      const double idx = 2 * q + 1 - r - c;
      const double v = 1.0 / (idx * fact(q - r) * fact(q - c));
      pc.Qt[r][c] = v;
      pc.Qt[c][r] = v;
    }
  // QLt = cholesky(Qt).L   (chol(Qt (x) I) == chol(Qt) (x) I)
  for (int j = 0; j < nb; ++j) {
    double s = pc.Qt[j][j];
    for (int k = 0; k < j; ++k) s -= pc.QLt[j][k] * pc.QLt[j][k];
    pc.QLt[j][j] = std::sqrt(s);
    for (int i = j + 1; i < nb; ++i) {
      double t = pc.Qt[i][j];
      for (int k = 0; k < j; ++k) t -= pc.QLt[i][k] * pc.QLt[j][k];
      pc.QLt[i][j] = t / pc.QLt[j][j];
    }
  }
}

size_t field_elem(int field) {
  switch (field) {
    case ODEF_F_NACCEPT: case ODEF_F_NREJECT: case ODEF_F_NF: case ODEF_F_NJAC: case ODEF_F_NSAVED: case ODEF_F_RETCODE:
      return sizeof(int32_t);
    default: return sizeof(double);
  }
}

// number of elements of a field for the current solve shape
size_t field_count(const odef_ctx* c, int field, long n_save) {
  const size_t N = (size_t)c->cfg.n_traj;
  switch (field) {
    case ODEF_F_MEAN: case ODEF_F_SMOOTH_MEAN: return (size_t)n_save * c->D * N;
    case ODEF_F_COV_TRIL: case ODEF_F_SMOOTH_COV_TRIL: return (size_t)n_save * c->TRI * N;
    case ODEF_F_DIFFUSION: return (size_t)n_save * N;
    case ODEF_F_T: return c->adaptive ? (size_t)n_save * N : (size_t)n_save;
    case ODEF_F_U0: return (size_t)c->d * N;
    case ODEF_F_DENSE_MEAN: return (size_t)c->n_q * c->D * N;
    case ODEF_F_DENSE_COV_TRIL: return (size_t)c->n_q * c->TRI * N;
    case ODEF_F_SAMPLES: return (size_t)n_save * c->D * (size_t)c->n_samples * N;
    default: return N;
  }
}

int ensure(odef_ctx* c, int field, size_t bytes) {
  Buf& b = c->f[field];
  if (b.bound) {
    if (b.bytes < bytes) return fail(c, "bound buffer for field %d too small: %zu < %zu bytes", field, b.bytes, bytes);
    b.valid = bytes;
    return 0;
  }
  if (b.bytes < bytes) {
    if (b.ptr) HIPCHK(c, hipFree(b.ptr));
    b.ptr = nullptr;
    b.bytes = 0;
    HIPCHK(c, hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    b.owned = true;
  }
  b.valid = bytes;
  return 0;
}

int ensure_ws(odef_ctx* c, size_t doubles) {
  if (c->ws_cap >= doubles) return 0;
  if (c->d_ws) HIPCHK(c, hipFree(c->d_ws));
  c->d_ws = nullptr;
  c->ws_cap = 0;
  HIPCHK(c, hipMalloc((void**)&c->d_ws, doubles * sizeof(double)));
  c->ws_cap = doubles;
  return 0;
}

// Stage for the covariance records of the workgroup-per-trajectory kernels: as many records as fit in `ODEF_SMOOTH_STAGE_MB`
// (default: a third of the free device memory), at most `want_recs`.  Returns the capacity in doubles (0: work in place).
size_t ensure_stage(odef_ctx* c, long want_recs, size_t per_rec_doubles) {
  if (want_recs < 2) return 0;
  size_t budget;
  if (const char* e = getenv("ODEF_SMOOTH_STAGE_MB")) {
    budget = (size_t)atol(e) << 20;
  } else {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    budget = (free_b + c->stage_cap * sizeof(double)) / 3;
  }
  size_t recs = budget / (per_rec_doubles * sizeof(double));
  if (recs > (size_t)want_recs) recs = (size_t)want_recs;
  if (recs < 2) return 0;
  const size_t doubles = recs * per_rec_doubles;
  if (c->stage_cap < doubles) {
    if (c->d_stage) (void)hipFree(c->d_stage);
    c->d_stage = nullptr;
    c->stage_cap = 0;
    if (hipMalloc((void**)&c->d_stage, doubles * sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    c->stage_cap = doubles;
  }
  return doubles;
}

int set_device(odef_ctx* c) {
  HIPCHK(c, hipSetDevice(c->device));
  return 0;
}

__global__ void transpose_in_kernel(const double* __restrict__ src /*[N][k]*/, double* __restrict__ dst /*[k][N]*/,
                                    long N, int k) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int a = 0; a < k; ++a) dst[(size_t)a * N + i] = src[(size_t)i * k + a];
}

__device__ inline unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct PerturbArgs { double base[32]; };

__global__ void perturbed_u0_kernel(PerturbArgs a, double* __restrict__ dst /*[d][N]*/, long N, int d, int n_pert,
                                    double scale, unsigned long long seed, long first) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int k = 0; k < d; ++k) {
    double v = a.base[k];
    if (k < n_pert) {
#pragma clang fp contract(off)
      const unsigned long long r = splitmix64(seed + (unsigned long long)n_pert * (unsigned long long)(first + i) + k);
      const double U = (double)(r >> 11) * 0x1.0p-53;
      // separate roundings (no FMA contraction) so that the ensemble is bit-identical to the host generators
      const double w = 2.0 * U - 1.0;
      const double sw = scale * w;
      v = a.base[k] + sw;
    }
    dst[(size_t)k * N + i] = v;
  }
}

__global__ void scale_cov_kernel(double* __restrict__ cov, double* __restrict__ diff, double* __restrict__ loglik,
                                 const int* __restrict__ nsaved, long N, long n_save_fixed, int TRI) {
  // postamble! for static diffusion (src/integrator_utils.jl:4-18): Sigma *= final_diff, diffusions .= final_diff,
  // sol.log_likelihood = NaN.  nsaved != nullptr: adaptive solve, per-trajectory record count.
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const long n_save = nsaved ? (long)nsaved[i] : n_save_fixed;
  const bool final_only = !nsaved && n_save_fixed == 1;  // final-save mode: slot 0 holds the last state and the last global diffusion
  if (!final_only && n_save < 2) {  // no step taken: sol.diffusions is empty, nothing to rescale
    loglik[i] = __builtin_nan("");
    return;
  }
  const double s = diff[(size_t)(n_save - 1) * N + i];
  for (long n = 0; n < n_save; ++n) {
    for (int k = 0; k < TRI; ++k) cov[((size_t)n * TRI + k) * N + i] *= s;
    if (n >= 1) diff[(size_t)n * N + i] = s;
  }
  loglik[i] = __builtin_nan("");
}

}  // namespace

extern "C" {

int odef_version(void) { return ODEF_VERSION; }

const char* odef_last_error(const odef_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int odef_rhs_compile(const char* name, const char* source, int32_t d, int32_t n_params, const char* include_dir, int32_t* rhs_id) {
  if (!rhs_id) return fail(nullptr, "odef_rhs_compile: null rhs_id");
  std::string err;
  const int id = jit_register(name, source, d, n_params, include_dir, err);
  if (id < 0) {
    g_create_error = err;  // the whole compiler log, not truncated
    return -1;
  }
  *rhs_id = id;
  return 0;
}

int odef_create(odef_ctx** out, const odef_config* cfg) {
  if (!out || !cfg) return fail(nullptr, "odef_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(odef_config))
    return fail(nullptr, "odef_create: struct_size %d != %zu", cfg->struct_size, sizeof(odef_config));
  RhsInfo ri{0, 0};
  if (cfg->rhs_id >= kJitFirstId) {
    if (!jit_lookup(cfg->rhs_id, &ri.d, &ri.np)) return fail(nullptr, "odef_create: unknown run-time rhs_id %d", cfg->rhs_id);
  } else {
    if (cfg->rhs_id < 0 || cfg->rhs_id > ODEF_RHS_LORENZ96) return fail(nullptr, "odef_create: unknown rhs_id %d", cfg->rhs_id);
    ri = kRhs[cfg->rhs_id];
  }
  if (cfg->d != ri.d) return fail(nullptr, "odef_create: rhs %d has dimension %d, got d=%d", cfg->rhs_id, ri.d, cfg->d);
  if (cfg->n_params != ri.np) return fail(nullptr, "odef_create: rhs %d has %d parameters, got %d", cfg->rhs_id, ri.np, cfg->n_params);
  if (cfg->order < 1 || cfg->order > ODEF_MAX_ORDER) return fail(nullptr, "odef_create: order %d outside 1..%d", cfg->order, ODEF_MAX_ORDER);
  if (cfg->alg != ODEF_EK0 && cfg->alg != ODEF_EK1) return fail(nullptr, "odef_create: unknown alg %d", cfg->alg);
  if (cfg->diffusion != ODEF_DIFFUSION_DYNAMIC && cfg->diffusion != ODEF_DIFFUSION_FIXED && cfg->diffusion != ODEF_DIFFUSION_FIXED_MAP)
    return fail(nullptr, "odef_create: unknown diffusion model %d", cfg->diffusion);
  if (cfg->n_traj <= 0) return fail(nullptr, "odef_create: n_traj must be positive");
  // run-time compiled fields: lane / row-team kernels up to state dimension 20 (d <= 10), the workgroup-per-trajectory kernels above
  const bool jit_team = cfg->rhs_id >= kJitFirstId && (cfg->d * (cfg->order + 1) > 20 || cfg->d > 10);
  if (jit_team && (cfg->d % 2 != 0 || cfg->d > 32 || cfg->d * (cfg->order + 1) > 176))
    return fail(nullptr, "odef_create: run-time compiled vector fields above state dimension 20 run on the workgroup-per-trajectory kernels: even d <= 32 and d(q+1) <= 176 (got d = %d, d(q+1) = %d)",
                cfg->d, cfg->d * (cfg->order + 1));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, "odef_create: no HIP device available (libodefilter_hip has no CPU path)");
  odef_ctx* c = new odef_ctx();
  c->cfg = *cfg;
  if (c->cfg.smooth) c->cfg.save_mode = ODEF_SAVE_EVERYSTEP;
  c->d = cfg->d;
  c->q = cfg->order;
  c->D = c->d * (c->q + 1);
  c->TRI = c->D * (c->D + 1) / 2;
  c->np = cfg->n_params;
  if (cfg->device >= 0) c->device = cfg->device;
  else if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
  if (c->device >= ndev) { delete c; return fail(nullptr, "odef_create: device %d not present (%d devices)", cfg->device, ndev); }
  build_prior(c->q, c->pc);
  c->team = team_launch(cfg->rhs_id);
  c->team_path = c->team != nullptr;
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess && jit_team) {
    std::string jerr;
    c->team = jit_get_team(cfg->rhs_id, c->q, cfg->alg == ODEF_EK1, team_abi_stamp(), jerr);
    c->team_path = c->team != nullptr;
    if (!c->team) {
      g_create_error = "odef_create: " + jerr;  // the whole compiler log
      delete c;
      return -1;
    }
  } else if (e == hipSuccess && cfg->rhs_id >= kJitFirstId) {
    std::string jerr;
    c->jit = jit_get_module(cfg->rhs_id, c->q, cfg->alg == ODEF_EK1, c->device, jerr);
    if (!c->jit) {
      fail(nullptr, "odef_create: %s", jerr.substr(0, 400).c_str());
      delete c;
      return -1;
    }
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreate(&c->ev[k]);
  const size_t N = (size_t)cfg->n_traj;
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_u0, sizeof(double) * c->d * N);
  if (e == hipSuccess && c->np > 0) e = hipMalloc((void**)&c->d_p, sizeof(double) * c->np * (cfg->params_shared ? 1 : N));
  if (e != hipSuccess) {
    fail(nullptr, "odef_create: HIP setup failed: %s", hipGetErrorString(e));
    odef_destroy(c);
    return -1;
  }
  c->stream = c->own_stream;
  c->f[ODEF_F_U0].ptr = c->d_u0;
  c->f[ODEF_F_U0].bytes = c->f[ODEF_F_U0].valid = sizeof(double) * c->d * N;
  *out = c;
  return 0;
}

void odef_destroy(odef_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (int k = 0; k < ODEF_F_COUNT_; ++k)
    if (k != ODEF_F_U0 && c->f[k].owned && c->f[k].ptr) (void)hipFree(c->f[k].ptr);
  if (c->d_u0) (void)hipFree(c->d_u0);
  if (c->d_p) (void)hipFree(c->d_p);
  if (c->d_ws) (void)hipFree(c->d_ws);
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->d_hs) (void)hipFree(c->d_hs);
  if (c->d_ptab) (void)hipFree(c->d_ptab);
  if (c->d_tgrid) (void)hipFree(c->d_tgrid);
  if (c->d_tq) (void)hipFree(c->d_tq);
  if (c->d_tab_idx) (void)hipFree(c->d_tab_idx);
  for (int k = 0; k < 4; ++k)
    if (c->ev[k]) (void)hipEventDestroy(c->ev[k]);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int odef_set_stream(odef_ctx* c, void* hip_stream) {
  if (!c) return -1;
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return 0;
}

int odef_set_problem(odef_ctx* c, const double* u0, const double* p, double t0) {
  if (!c || !u0) return fail(c, "odef_set_problem: null argument");
  if (c->np > 0 && !p) return fail(c, "odef_set_problem: parameters required");
  if (set_device(c)) return -1;
  const long N = c->cfg.n_traj;
  double* tmp = nullptr;
  const size_t ub = sizeof(double) * c->d * N;
  const size_t pb = (c->np > 0 && !c->cfg.params_shared) ? sizeof(double) * c->np * N : 0;
  HIPCHK(c, hipMalloc((void**)&tmp, ub > pb ? ub : pb));
  HIPCHK(c, hipMemcpyAsync(tmp, u0, ub, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(transpose_in_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, tmp, c->d_u0, N, c->d);
  if (c->np > 0) {
    if (c->cfg.params_shared) {
      HIPCHK(c, hipMemcpyAsync(c->d_p, p, sizeof(double) * c->np, hipMemcpyHostToDevice, c->stream));
    } else {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      HIPCHK(c, hipMemcpyAsync(tmp, p, pb, hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(transpose_in_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, tmp, c->d_p, N, c->np);
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipFree(tmp));
  c->t0 = t0;
  c->have_problem = true;
  return 0;
}

int odef_set_problem_device(odef_ctx* c, const double* d_u0, const double* d_p, double t0) {
  if (!c || !d_u0) return fail(c, "odef_set_problem_device: null argument");
  if (c->np > 0 && !d_p) return fail(c, "odef_set_problem_device: parameters required");
  if (set_device(c)) return -1;
  const size_t N = (size_t)c->cfg.n_traj;
  HIPCHK(c, hipMemcpyAsync(c->d_u0, d_u0, sizeof(double) * c->d * N, hipMemcpyDeviceToDevice, c->stream));
  if (c->np > 0)
    HIPCHK(c, hipMemcpyAsync(c->d_p, d_p, sizeof(double) * c->np * (c->cfg.params_shared ? 1 : N), hipMemcpyDeviceToDevice, c->stream));
  c->t0 = t0;
  c->have_problem = true;
  return 0;
}

int odef_set_problem_perturbed(odef_ctx* c, const double* base_u0, const double* p, double t0, double scale,
                               uint64_t seed, int64_t first_index, int32_t n_perturbed) {
  if (!c || !base_u0) return fail(c, "odef_set_problem_perturbed: null argument");
  if (c->np > 0 && !p) return fail(c, "odef_set_problem_perturbed: parameters required");
  if (!c->cfg.params_shared && c->np > 0) return fail(c, "odef_set_problem_perturbed: needs params_shared = 1");
  if (n_perturbed < 0 || n_perturbed > c->d) return fail(c, "odef_set_problem_perturbed: n_perturbed %d outside 0..%d", n_perturbed, c->d);
  if (set_device(c)) return -1;
  PerturbArgs a;
  for (int k = 0; k < c->d; ++k) a.base[k] = base_u0[k];
  const long N = c->cfg.n_traj;
  hipLaunchKernelGGL(perturbed_u0_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, a, c->d_u0, N, c->d,
                     (int)n_perturbed, scale, (unsigned long long)seed, (long)first_index);
  HIPCHK(c, hipGetLastError());
  if (c->np > 0) HIPCHK(c, hipMemcpyAsync(c->d_p, p, sizeof(double) * c->np, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->t0 = t0;
  c->have_problem = true;
  return 0;
}

static int alloc_outputs(odef_ctx* c, long n_save) {
  static const int per_save[] = {ODEF_F_MEAN, ODEF_F_COV_TRIL, ODEF_F_DIFFUSION};
  for (int f : per_save)
    if (ensure(c, f, field_count(c, f, n_save) * sizeof(double))) return -1;
  if (c->adaptive && ensure(c, ODEF_F_T, field_count(c, ODEF_F_T, n_save) * sizeof(double))) return -1;
  static const int per_traj[] = {ODEF_F_LOGLIK, ODEF_F_NACCEPT, ODEF_F_NREJECT, ODEF_F_NF, ODEF_F_NJAC, ODEF_F_NSAVED, ODEF_F_RETCODE};
  for (int f : per_traj)
    if (ensure(c, f, (size_t)c->cfg.n_traj * field_elem(f))) return -1;
  return 0;
}

static void fill_params(odef_ctx* c, FilterParams& P) {
  std::memset(&P, 0, sizeof P);
  P.pc = c->pc;
  P.u0 = c->d_u0;
  P.p = c->d_p;
  P.p_shared = c->cfg.params_shared;
  P.N = c->cfg.n_traj;
  P.everystep = c->cfg.save_mode == ODEF_SAVE_EVERYSTEP;
  P.fixed_diffusion = (int)c->cfg.diffusion;  // 0 dynamic, 1 fixed, 2 fixedMAP (static_diffusion_update, ek_math.h)
  P.want_loglik = c->cfg.want_loglik;
  {
    const char* e = getenv("ODEF_STAGGER");
    P.stagger = e ? atoi(e) : 0;
  }
  P.mean = (double*)c->f[ODEF_F_MEAN].ptr;
  P.cov = (double*)c->f[ODEF_F_COV_TRIL].ptr;
  P.diff = (double*)c->f[ODEF_F_DIFFUSION].ptr;
  P.tsave = (double*)c->f[ODEF_F_T].ptr;
  P.loglik = (double*)c->f[ODEF_F_LOGLIK].ptr;
  P.naccept = (int*)c->f[ODEF_F_NACCEPT].ptr;
  P.nreject = (int*)c->f[ODEF_F_NREJECT].ptr;
  P.nf = (int*)c->f[ODEF_F_NF].ptr;
  P.njac = (int*)c->f[ODEF_F_NJAC].ptr;
  P.nsaved = (int*)c->f[ODEF_F_NSAVED].ptr;
  P.retcode = (int*)c->f[ODEF_F_RETCODE].ptr;
}

static int complete_pending(odef_ctx* c) {
  if (c->pending == 1) {
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    HIPCHK(c, hipEventElapsedTime(&c->ms[0], c->ev[0], c->ev[1]));
  } else if (c->pending == 2) {
    HIPCHK(c, hipEventSynchronize(c->ev[3]));
    HIPCHK(c, hipEventElapsedTime(&c->ms[1], c->ev[2], c->ev[3]));
  }
  c->pending = 0;
  return 0;
}

static int finish_filter(odef_ctx* c, int nlaunch) {
  // static diffusion: rescale all covariances by the final global diffusion (src/integrator_utils.jl:4-18)
  if (c->cfg.diffusion != ODEF_DIFFUSION_DYNAMIC) {
    const long N = c->cfg.n_traj;
    hipLaunchKernelGGL(scale_cov_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, c->stream,
                       (double*)c->f[ODEF_F_COV_TRIL].ptr, (double*)c->f[ODEF_F_DIFFUSION].ptr,
                       (double*)c->f[ODEF_F_LOGLIK].ptr, c->adaptive ? (const int*)c->f[ODEF_F_NSAVED].ptr : (const int*)nullptr,
                       N, c->n_save, c->TRI);
    ++nlaunch;
    c->stage_filter_recs = 0;  // (records the filter may have left in the stage are the unscaled ones)
  }
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  HIPCHK(c, hipGetLastError());
  c->nl[0] = nlaunch;
  c->solved = true;
  c->smoothed_done = false;
  c->pending = 1;
  return c->defer ? 0 : complete_pending(c);
}

int odef_solve_fixed(odef_ctx* c, const double* tgrid, int64_t n_t) {
  if (!c || !tgrid) return fail(c, "odef_solve_fixed: null argument");
  if (!c->have_problem) return fail(c, "odef_solve_fixed: call odef_set_problem first");
  if (n_t < 2) return fail(c, "odef_solve_fixed: need at least two grid points (adaptive=false requires a dt)");
  if (tgrid[0] != c->t0) return fail(c, "odef_solve_fixed: tgrid[0] = %g differs from t0 = %g", tgrid[0], c->t0);
  for (int64_t n = 0; n + 1 < n_t; ++n)
    if (!(tgrid[n + 1] > tgrid[n])) return fail(c, "odef_solve_fixed: tgrid must be strictly increasing (index %lld)", (long long)n);
  if (set_device(c)) return -1;
  const long nsteps = (long)n_t - 1;
  c->adaptive = false;
  c->n_save = (c->cfg.save_mode == ODEF_SAVE_EVERYSTEP) ? nsteps + 1 : 1;
  // a repeated solve on the grid that is already on the device (ensemble sweeps, the benchmark loop) skips the table
  // construction, the four uploads and their synchronisation
  const bool same_grid = c->grid_on_device && c->tgrid.size() == (size_t)n_t &&
                         std::memcmp(c->tgrid.data(), tgrid, sizeof(double) * (size_t)n_t) == 0;
  if (!same_grid) {
    c->grid_on_device = false;
    c->tgrid.assign(tgrid, tgrid + n_t);
    // step sizes, and one preconditioner table per distinct h (src/preconditioning.jl:1-17; pval by libm pow
    // as the reference's h^(-q-1/2))
    std::vector<double> hs(nsteps), tabs;
    std::vector<int> idx(nsteps);
    std::vector<double> distinct;
    for (long n = 0; n < nsteps; ++n) {
      hs[n] = tgrid[n + 1] - tgrid[n];
      int k = -1;
      for (size_t j = distinct.size(); j-- > 0;)
        if (distinct[j] == hs[n]) { k = (int)j; break; }
      if (k < 0) {
        k = (int)distinct.size();
        distinct.push_back(hs[n]);
        tabs.resize(distinct.size() * kTabStride, 0.0);
        double* t = tabs.data() + (size_t)k * kTabStride;
        const double pval = std::pow(hs[n], -c->q - 0.5);
        switch (c->q) {
          case 1: precond_fill<2>(hs[n], pval, t); break;
          case 2: precond_fill<3>(hs[n], pval, t); break;
          case 3: precond_fill<4>(hs[n], pval, t); break;
          case 4: precond_fill<5>(hs[n], pval, t); break;
          default: precond_fill<6>(hs[n], pval, t); break;
        }
      }
      idx[n] = k;
    }
    if (c->grid_cap < (size_t)n_t) {
      if (c->d_hs) { HIPCHK(c, hipFree(c->d_hs)); HIPCHK(c, hipFree(c->d_ptab)); HIPCHK(c, hipFree(c->d_tab_idx)); HIPCHK(c, hipFree(c->d_tgrid)); }
      c->d_hs = c->d_ptab = c->d_tgrid = nullptr;
      c->d_tab_idx = nullptr;
      c->grid_cap = 0;
      HIPCHK(c, hipMalloc((void**)&c->d_hs, sizeof(double) * n_t));
      HIPCHK(c, hipMalloc((void**)&c->d_ptab, sizeof(double) * n_t * kTabStride));  // worst case: all h distinct
      HIPCHK(c, hipMalloc((void**)&c->d_tab_idx, sizeof(int) * n_t));
      HIPCHK(c, hipMalloc((void**)&c->d_tgrid, sizeof(double) * n_t));
      c->grid_cap = (size_t)n_t;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_hs, hs.data(), sizeof(double) * nsteps, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_ptab, tabs.data(), sizeof(double) * tabs.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_tab_idx, idx.data(), sizeof(int) * nsteps, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_tgrid, tgrid, sizeof(double) * n_t, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the host vectors are locals
    c->grid_on_device = true;
  }
  if (alloc_outputs(c, c->n_save)) return -1;
  FilterParams P;
  fill_params(c, P);
  P.hs = c->d_hs;
  P.ptab = c->d_ptab;
  P.tab_idx = c->d_tab_idx;
  P.nsteps = nsteps;
  P.t0 = c->t0;
  int rc;
  if (c->team_path) {
    size_t have = 0;
    if (P.everystep) {  // the matrix-core kernel writes its records through the trajectory-major stage when all of them fit
      const size_t tri = (size_t)c->D * (c->D + 1) / 2, per_rec = (size_t)P.N * ((tri + 15) / 16 * 16);
      have = ensure_stage(c, nsteps + 1, per_rec);
      if (have < (size_t)(nsteps + 1) * per_rec) have = 0;
    }
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    rc = c->team->filter(c->q, c->cfg.alg == ODEF_EK1, P, c->stream, 0, have ? c->d_stage : nullptr, have, &c->stage_filter_recs);
  } else {
    // one lane per trajectory: the per-field buffer descriptors carry 32-bit sizes
    if ((size_t)c->TRI * (size_t)c->cfg.n_traj * sizeof(double) >= (1ull << 31))
      return fail(c, "odef_solve_fixed: n_traj * D(D+1)/2 * 8 bytes must stay below 2 GiB per save slot; shard the ensemble");
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if (c->jit && c->jit->rows16 && P.N < filter_rows_max_n()) {  // small ensemble: 16 lanes per trajectory, as for the compiled-in fields
      note_kernel(P.everystep ? "odef_jit_rows_fixed_every" : "odef_jit_rows_fixed_final");
      rc = jit_launch(P.everystep ? c->jit->rows_fixed_every : c->jit->rows_fixed_final, rows_grid(P.N), 1, &P, c->stream, 256);
    } else if (c->jit) {
      note_kernel(P.everystep ? "odef_jit_fixed_every" : "odef_jit_fixed_final");
      rc = jit_launch(P.everystep ? c->jit->fixed_every : c->jit->fixed_final, (unsigned)((P.N + 63) / 64), 1, &P, c->stream);
    } else
      rc = launch_filter(c->cfg.rhs_id, c->q, c->cfg.alg == ODEF_EK1, 0, P, c->stream);
  }
  if (rc) return fail(c, "odef_solve_fixed: no kernel for rhs %d order %d", c->cfg.rhs_id, c->q);
  std::snprintf(c->kname[0], sizeof c->kname[0], "%s", last_kernel());
  return finish_filter(c, 1);
}

int odef_solve_adaptive(odef_ctx* c, double t1, double abstol, double reltol, double dt0, const odef_controller* ctrl,
                        int64_t max_steps) {
  if (!c) return -1;
  if (!c->have_problem) return fail(c, "odef_solve_adaptive: call odef_set_problem first");
  if (!(t1 > c->t0)) return fail(c, "odef_solve_adaptive: t1 must exceed t0");
  if (!(dt0 > 0.0)) return fail(c, "odef_solve_adaptive: dt0 must be positive");
  if (max_steps < 1) return fail(c, "odef_solve_adaptive: max_steps must be >= 1");
  if (c->cfg.save_mode != ODEF_SAVE_EVERYSTEP) return fail(c, "odef_solve_adaptive: needs ODEF_SAVE_EVERYSTEP");
  if ((size_t)c->TRI * (size_t)c->cfg.n_traj * sizeof(double) >= (1ull << 31))
    return fail(c, "odef_solve_adaptive: n_traj * D(D+1)/2 * 8 bytes must stay below 2 GiB per save slot; shard the ensemble");
  if (set_device(c)) return -1;
  c->adaptive = true;
  c->n_save = (long)max_steps + 1;
  c->tgrid.clear();
  if (alloc_outputs(c, c->n_save)) return -1;
  FilterParams P;
  fill_params(c, P);
  P.t0 = c->t0;
  P.t1 = t1;
  P.abstol = abstol;
  P.reltol = reltol;
  P.dt0 = dt0;
  P.max_save = c->n_save;
  if (ctrl) {
    std::memcpy(&P.ctrl, ctrl, sizeof(Controller));
  } else {  // OrdinaryDiffEq defaults with src/alg_utils.jl:23-24 exponents
    P.ctrl = Controller{7.0 / (10.0 * (c->q + 1)), 2.0 / (5.0 * (c->q + 1)), 0.9, 0.2, 10.0, 1.0, 1.0, 1e-4, 0.0, 1e300};
  }
  HIPCHK(c, hipMemsetAsync(c->f[ODEF_F_T].ptr, 0, c->f[ODEF_F_T].valid, c->stream));
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  int rc;
  if (c->jit && c->jit->rows16 && P.N < filter_rows_max_n()) {
    note_kernel("odef_jit_rows_adaptive");
    rc = jit_launch(c->jit->rows_adaptive, rows_grid(P.N), 1, &P, c->stream, 256);
  } else if (c->jit) {
    note_kernel("odef_jit_adaptive");
    rc = jit_launch(c->jit->adaptive, (unsigned)((P.N + 63) / 64), 1, &P, c->stream);
  } else {
    rc = c->team_path ? c->team->filter(c->q, c->cfg.alg == ODEF_EK1, P, c->stream, 1, nullptr, 0, &c->stage_filter_recs)
                      : launch_filter(c->cfg.rhs_id, c->q, c->cfg.alg == ODEF_EK1, 1, P, c->stream);
  }
  if (rc) return fail(c, "odef_solve_adaptive: no kernel for rhs %d order %d", c->cfg.rhs_id, c->q);
  std::snprintf(c->kname[0], sizeof c->kname[0], "%s", last_kernel());
  return finish_filter(c, 1);
}

int odef_smooth(odef_ctx* c) {
  if (!c) return -1;
  if (!c->solved) return fail(c, "odef_smooth: nothing to smooth, call odef_solve_* first");
  if (c->cfg.save_mode != ODEF_SAVE_EVERYSTEP) return fail(c, "odef_smooth: needs ODEF_SAVE_EVERYSTEP (cfg.smooth = 1)");
  if (set_device(c)) return -1;
  if (ensure(c, ODEF_F_SMOOTH_MEAN, field_count(c, ODEF_F_SMOOTH_MEAN, c->n_save) * sizeof(double))) return -1;
  if (ensure(c, ODEF_F_SMOOTH_COV_TRIL, field_count(c, ODEF_F_SMOOTH_COV_TRIL, c->n_save) * sizeof(double))) return -1;
  SmoothParams S;
  std::memset(&S, 0, sizeof S);
  S.pc = c->pc;
  S.N = c->cfg.n_traj;
  S.n_save = c->n_save;
  S.adaptive = c->adaptive;
  S.hs = c->d_hs;
  S.ptab = c->d_ptab;
  S.tab_idx = c->d_tab_idx;
  S.tsave = (const double*)c->f[ODEF_F_T].ptr;
  S.nsaved = (const int*)c->f[ODEF_F_NSAVED].ptr;
  S.mean = (const double*)c->f[ODEF_F_MEAN].ptr;
  S.cov = (const double*)c->f[ODEF_F_COV_TRIL].ptr;
  S.diff = (const double*)c->f[ODEF_F_DIFFUSION].ptr;
  S.smean = (double*)c->f[ODEF_F_SMOOTH_MEAN].ptr;
  S.scov = (double*)c->f[ODEF_F_SMOOTH_COV_TRIL].ptr;
  S.retcode = (int*)c->f[ODEF_F_RETCODE].ptr;
  if (c->adaptive) {  // unused save slots stay defined
    HIPCHK(c, hipMemsetAsync(S.smean, 0, c->f[ODEF_F_SMOOTH_MEAN].valid, c->stream));
    HIPCHK(c, hipMemsetAsync(S.scov, 0, c->f[ODEF_F_SMOOTH_COV_TRIL].valid, c->stream));
  }
  if (c->team_path && ensure_ws(c, (size_t)c->cfg.n_traj * c->team->smooth_ws(c->q))) return -1;
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  int rc;
  if (c->jit)
    if (c->jit->rows16 && S.N < smooth_rows_max_n()) {  // small ensemble: the DPP row-team smoother
      note_kernel(S.adaptive ? "odef_jit_bcast_adapt" : "odef_jit_bcast_fixed");
      rc = jit_launch(S.adaptive ? c->jit->bcast_adapt : c->jit->bcast_fixed, rows_grid(S.N), 1, &S, c->stream, 256);
    } else if (c->jit->posterior) {
      note_kernel(S.adaptive ? "odef_jit_smooth_adapt" : "odef_jit_smooth_fixed");
      rc = jit_launch(S.adaptive ? c->jit->smooth_adapt : c->jit->smooth_fixed, (unsigned)((S.N + 63) / 64), 1, &S, c->stream);
    } else if (c->jit->smooth_rows) {
      const long tpb = 64 / c->jit->rows_team;
      note_kernel("odef_jit_smooth_rows");
      rc = jit_launch(c->jit->smooth_rows, (unsigned)((S.N + tpb - 1) / tpb), 1, &S, c->stream);
    } else
      rc = -3;
  else if (c->team_path) {
    rc = -4;
    {  // the covariance records go through the trajectory-major stage, in blocks (record_stage.h)
      long n_rec = S.n_save;
      if (S.adaptive) {  // save slots in use: the largest record count of the ensemble
        std::vector<int> ns((size_t)S.N);
        HIPCHK(c, hipMemcpyAsync(ns.data(), S.nsaved, ns.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        n_rec = 1;
        for (int v : ns) n_rec = v > n_rec ? v : n_rec;
        if (n_rec > S.n_save) n_rec = S.n_save;
      }
      const size_t tri = (size_t)c->D * (c->D + 1) / 2;
      const size_t per_rec = (size_t)S.N * ((tri + 15) / 16 * 16);
      // the filter's records may still be in the stage (fixed grid, every step saved, all of them fit): record r at r * per_rec
      const bool resident = !S.adaptive && n_rec >= 3 && c->stage_filter_recs == n_rec && c->stage_cap >= (size_t)n_rec * per_rec;
      const size_t have = n_rec < 3 ? 0 : resident ? (size_t)(n_rec - 1) * per_rec : ensure_stage(c, n_rec - 1, per_rec);
      if (have) rc = c->team->smooth_staged(c->q, S, n_rec, c->d_ws, c->d_stage, have, c->stream, resident ? n_rec : 0);
      c->stage_filter_recs = 0;  // (smoothed records now, or another block layout)
    }
    if (rc == -4) rc = c->team->smooth(c->q, S, c->d_ws, c->stream);
  } else
    rc = launch_smooth(c->d, c->q, S, c->stream);
  if (rc) return fail(c, "odef_smooth: no kernel for d %d order %d", c->d, c->q);
  std::snprintf(c->kname[1], sizeof c->kname[1], "%s", last_kernel());
  HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
  HIPCHK(c, hipGetLastError());
  c->nl[1] = 1;
  c->smoothed_done = true;
  c->pending = 2;
  return c->defer ? 0 : complete_pending(c);
}

int odef_dense_output(odef_ctx* c, const double* tq, int64_t n_q, int smoothed) {
  if (!c || !tq) return fail(c, "odef_dense_output: null argument");
  if (!c->solved) return fail(c, "odef_dense_output: call odef_solve_* first");
  if (c->cfg.save_mode != ODEF_SAVE_EVERYSTEP) return fail(c, "odef_dense_output: needs ODEF_SAVE_EVERYSTEP");
  if (smoothed && !c->smoothed_done) return fail(c, "odef_dense_output: smoothed posterior requested but odef_smooth has not run");
  if (n_q < 1 || n_q > 65535) return fail(c, "odef_dense_output: n_q must be in 1..65535");
  if (!c->team_path && c->D > 32) return fail(c, "odef_dense_output: built for state dimension <= 32 and for the workgroup-per-trajectory path (got %d)", c->D);
  if (set_device(c)) return -1;
  if (c->tq_cap < (size_t)n_q) {
    if (c->d_tq) HIPCHK(c, hipFree(c->d_tq));
    c->d_tq = nullptr;
    HIPCHK(c, hipMalloc((void**)&c->d_tq, sizeof(double) * n_q));
    c->tq_cap = (size_t)n_q;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_tq, tq, sizeof(double) * n_q, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->n_q = (long)n_q;
  const size_t N = (size_t)c->cfg.n_traj;
  if (ensure(c, ODEF_F_DENSE_MEAN, (size_t)n_q * c->D * N * sizeof(double))) return -1;
  if (ensure(c, ODEF_F_DENSE_COV_TRIL, (size_t)n_q * c->TRI * N * sizeof(double))) return -1;
  DenseParams P;
  std::memset(&P, 0, sizeof P);
  P.pc = c->pc;
  P.N = c->cfg.n_traj;
  P.n_save = c->n_save;
  P.adaptive = c->adaptive;
  P.smoothed = smoothed != 0;
  P.tgrid = c->d_tgrid;
  P.tsave = (const double*)c->f[ODEF_F_T].ptr;
  P.nsaved = (const int*)c->f[ODEF_F_NSAVED].ptr;
  P.mean = (const double*)c->f[ODEF_F_MEAN].ptr;
  P.cov = (const double*)c->f[ODEF_F_COV_TRIL].ptr;
  P.diff = (const double*)c->f[ODEF_F_DIFFUSION].ptr;
  P.smean = (const double*)c->f[ODEF_F_SMOOTH_MEAN].ptr;
  P.scov = (const double*)c->f[ODEF_F_SMOOTH_COV_TRIL].ptr;
  P.tq = c->d_tq;
  P.n_q = (long)n_q;
  P.qmean = (double*)c->f[ODEF_F_DENSE_MEAN].ptr;
  P.qcov = (double*)c->f[ODEF_F_DENSE_COV_TRIL].ptr;
  if (c->team_path && ensure_ws(c, (size_t)dense_d28_grid(P.N * P.n_q) * c->team->smooth_ws(c->q))) return -1;
  const long rows_tpb = c->jit ? 64 / c->jit->rows_team : 1;
  const int rc = c->jit ? (c->jit->posterior    ? jit_launch(c->jit->dense, (unsigned)((P.N + 63) / 64), (unsigned)P.n_q, &P, c->stream)
                           : c->jit->dense_rows ? jit_launch(c->jit->dense_rows, (unsigned)((P.N * P.n_q + rows_tpb - 1) / rows_tpb), 1, &P, c->stream)
                                                : -3)
                 : c->team_path ? c->team->dense(c->q, P, c->d_ws, c->stream)
                 : c->d == 2 ? launch_dense_d2(c->q, P, c->stream) : c->d == 3 ? launch_dense_d3(c->q, P, c->stream) : -3;
  if (rc) return fail(c, "odef_dense_output: no kernel for d %d order %d", c->d, c->q);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int odef_sample(odef_ctx* c, int64_t n_samples, uint64_t seed, double noise_scale) {
  if (!c) return -1;
  if (!c->solved) return fail(c, "odef_sample: call odef_solve_* first");
  if (c->cfg.save_mode != ODEF_SAVE_EVERYSTEP) return fail(c, "odef_sample: needs ODEF_SAVE_EVERYSTEP (sampling not implemented for non-smoothed posteriors)");
  if (n_samples < 1 || n_samples > 65535) return fail(c, "odef_sample: n_samples must be in 1..65535");
  if (!c->team_path && c->D > 32) return fail(c, "odef_sample: built for state dimension <= 32 and for the workgroup-per-trajectory path (got %d)", c->D);
  if (set_device(c)) return -1;
  c->n_samples = (long)n_samples;
  if (ensure(c, ODEF_F_SAMPLES, field_count(c, ODEF_F_SAMPLES, c->n_save) * sizeof(double))) return -1;
  SampleParams S;
  std::memset(&S, 0, sizeof S);
  S.pc = c->pc;
  S.N = c->cfg.n_traj;
  S.n_save = c->n_save;
  S.adaptive = c->adaptive;
  S.hs = c->d_hs;
  S.ptab = c->d_ptab;
  S.tab_idx = c->d_tab_idx;
  S.tsave = (const double*)c->f[ODEF_F_T].ptr;
  S.nsaved = (const int*)c->f[ODEF_F_NSAVED].ptr;
  S.mean = (const double*)c->f[ODEF_F_MEAN].ptr;
  S.cov = (const double*)c->f[ODEF_F_COV_TRIL].ptr;
  S.diff = (const double*)c->f[ODEF_F_DIFFUSION].ptr;
  S.n_samples = (long)n_samples;
  S.seed = (unsigned long long)seed;
  S.noise_scale = noise_scale;
  S.samples = (double*)c->f[ODEF_F_SAMPLES].ptr;
  if (c->adaptive) HIPCHK(c, hipMemsetAsync(S.samples, 0, c->f[ODEF_F_SAMPLES].valid, c->stream));  // unused slots stay defined
  if (c->team_path && ensure_ws(c, (size_t)dense_d28_grid(S.N * S.n_samples) * c->team->smooth_ws(c->q))) return -1;
  const long rows_tpb = c->jit ? 64 / c->jit->rows_team : 1;
  const int rc = c->jit ? (c->jit->posterior     ? jit_launch(c->jit->sample, (unsigned)((S.N + 63) / 64), (unsigned)S.n_samples, &S, c->stream)
                           : c->jit->sample_rows ? jit_launch(c->jit->sample_rows, (unsigned)((S.N * S.n_samples + rows_tpb - 1) / rows_tpb), 1, &S, c->stream)
                                                 : -3)
                 : c->team_path ? c->team->sample(c->q, S, c->d_ws, c->stream)
                 : c->d == 2 ? launch_sample_d2(c->q, S, c->stream) : c->d == 3 ? launch_sample_d3(c->q, S, c->stream) : -3;
  if (rc) return fail(c, "odef_sample: no kernel for d %d order %d", c->d, c->q);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int odef_dense_sample(odef_ctx* c, const double* tq, int64_t n_q, int64_t n_samples, uint64_t seed, double noise_scale) {
  if (!c || !tq) return fail(c, "odef_dense_sample: null argument");
  if (n_samples < 1 || n_samples > 65535) return fail(c, "odef_dense_sample: n_samples must be in 1..65535");
  for (int64_t k = 1; k < n_q; ++k)
    if (!(tq[k] >= tq[k - 1])) return fail(c, "odef_dense_sample: times must be non-decreasing");
  if (odef_dense_output(c, tq, n_q, 0)) return -1;  // filter posterior at tq (src/solution_sampling.jl:66)
  c->n_samples = (long)n_samples;
  if (ensure(c, ODEF_F_SAMPLES, field_count(c, ODEF_F_SAMPLES, (long)n_q) * sizeof(double))) return -1;
  SampleParams S;
  std::memset(&S, 0, sizeof S);
  S.pc = c->pc;
  S.N = c->cfg.n_traj;
  S.n_save = (long)n_q;
  S.adaptive = c->adaptive;
  S.tsave = (const double*)c->f[ODEF_F_T].ptr;
  S.nsaved = (const int*)c->f[ODEF_F_NSAVED].ptr;
  S.mean = (const double*)c->f[ODEF_F_DENSE_MEAN].ptr;
  S.cov = (const double*)c->f[ODEF_F_DENSE_COV_TRIL].ptr;
  S.diff = (const double*)c->f[ODEF_F_DIFFUSION].ptr;
  S.tq = c->d_tq;
  S.rec_t = c->d_tgrid;
  S.n_rec = c->n_save;
  S.n_samples = (long)n_samples;
  S.seed = (unsigned long long)seed;
  S.noise_scale = noise_scale;
  S.samples = (double*)c->f[ODEF_F_SAMPLES].ptr;
  if (c->team_path && ensure_ws(c, (size_t)dense_d28_grid(S.N * S.n_samples) * c->team->smooth_ws(c->q))) return -1;
  const long rows_tpb = c->jit ? 64 / c->jit->rows_team : 1;
  const int rc = c->jit ? (c->jit->posterior     ? jit_launch(c->jit->sample, (unsigned)((S.N + 63) / 64), (unsigned)S.n_samples, &S, c->stream)
                           : c->jit->sample_rows ? jit_launch(c->jit->sample_rows, (unsigned)((S.N * S.n_samples + rows_tpb - 1) / rows_tpb), 1, &S, c->stream)
                                                 : -3)
                 : c->team_path ? c->team->sample(c->q, S, c->d_ws, c->stream)
                 : c->d == 2 ? launch_sample_d2(c->q, S, c->stream) : c->d == 3 ? launch_sample_d3(c->q, S, c->stream) : -3;
  if (rc) return fail(c, "odef_dense_sample: no kernel for d %d order %d", c->d, c->q);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int64_t odef_n_save(const odef_ctx* c) { return c ? c->n_save : -1; }

int odef_field_bytes(const odef_ctx* c, int field, size_t* bytes) {
  if (!c || !bytes || field < 0 || field >= ODEF_F_COUNT_) return -1;
  if (field == ODEF_F_T && !c->adaptive) { *bytes = c->tgrid.size() ? (size_t)c->n_save * sizeof(double) : 0; return 0; }
  *bytes = c->f[field].valid;
  return 0;
}

int odef_get(odef_ctx* c, int field, void* host_dst, size_t bytes) {
  if (!c || !host_dst) return fail(c, "odef_get: null argument");
  if (field < 0 || field >= ODEF_F_COUNT_) return fail(c, "odef_get: unknown field %d", field);
  if (set_device(c)) return -1;
  if (field == ODEF_F_T && !c->adaptive) {
    if (c->tgrid.empty()) return fail(c, "odef_get: no solve yet");
    const size_t need = (size_t)c->n_save * sizeof(double);
    if (bytes != need) return fail(c, "odef_get: field T holds %zu bytes, caller asked for %zu", need, bytes);
    if (c->n_save == 1) ((double*)host_dst)[0] = c->tgrid.back();
    else std::memcpy(host_dst, c->tgrid.data(), need);
    return 0;
  }
  const Buf& b = c->f[field];
  if (!b.ptr || !b.valid) return fail(c, "odef_get: field %d holds no data yet", field);
  if (bytes != b.valid) return fail(c, "odef_get: field %d holds %zu bytes, caller asked for %zu", field, b.valid, bytes);
  HIPCHK(c, hipMemcpyAsync(host_dst, b.ptr, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int odef_get_device(odef_ctx* c, int field, void** dev_ptr, size_t* bytes) {
  if (!c || !dev_ptr || !bytes) return fail(c, "odef_get_device: null argument");
  if (field < 0 || field >= ODEF_F_COUNT_) return fail(c, "odef_get_device: unknown field %d", field);
  const Buf& b = c->f[field];
  if (!b.ptr || !b.valid) return fail(c, "odef_get_device: field %d holds no data yet", field);
  *dev_ptr = b.ptr;
  *bytes = b.valid;
  return 0;
}

int odef_bind_device(odef_ctx* c, int field, void* dev_ptr, size_t bytes) {
  if (!c) return -1;
  if (field < 0 || field >= ODEF_F_COUNT_ || field == ODEF_F_U0) return fail(c, "odef_bind_device: field %d cannot be bound", field);
  if (set_device(c)) return -1;
  Buf& b = c->f[field];
  if (b.owned && b.ptr) HIPCHK(c, hipFree(b.ptr));
  b = Buf{};
  if (dev_ptr) {
    b.ptr = dev_ptr;
    b.bytes = bytes;
    b.bound = true;
  }
  return 0;
}

int odef_synchronize(odef_ctx* c) {
  if (!c) return -1;
  if (set_device(c)) return -1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int odef_kernel_name(odef_ctx* c, int which, char* buf, size_t n) {
  if (!c || which < 0 || which > 1 || !buf || n == 0) return -1;
  std::snprintf(buf, n, "%s", c->kname[which]);
  return 0;
}

int odef_kernel_time_ms(odef_ctx* c, int which, float* ms, int* n_launches) {
  if (!c || which < 0 || which > 1 || !ms) return -1;
  *ms = c->ms[which];
  if (n_launches) *n_launches = c->nl[which];
  return 0;
}

int odef_ibm(int d, int q, double* A, double* Q_L) {
  if (d < 1 || q < 1 || q > ODEF_MAX_ORDER || !A || !Q_L) return -1;
  PriorConsts pc;
  build_prior(q, pc);
  const int D = d * (q + 1);
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) {
      const bool same = (i % d) == (j % d);
      A[i * D + j] = same ? pc.At[i / d][j / d] : 0.0;
      Q_L[i * D + j] = same ? pc.QLt[i / d][j / d] : 0.0;
    }
  return 0;
}

int odef_preconditioner(int d, int q, double h, double* P_diag) {
  if (d < 1 || q < 0 || !P_diag || !(h > 0.0)) return -1;
  double val = std::pow(h, -q - 0.5);  // src/preconditioning.jl:9
  for (int j = 0; j <= q; ++j) {
    for (int i = 0; i < d; ++i) P_diag[j * d + i] = val;
    val *= h;
  }
  return 0;
}


/* ---- ensemble sharded over the GPUs of one node (single process) ------------------------------------------------- */

int odef_shard_range(int64_t n_traj, int32_t n_shards, int32_t shard, int64_t* first, int64_t* count) {
  if (n_traj < 0 || n_shards < 1 || shard < 0 || shard >= n_shards || !first || !count) return -1;
  // contiguous blocks, the first n_traj % n_shards shards one trajectory longer
  const int64_t base = n_traj / n_shards, rem = n_traj % n_shards;
  *first = shard * base + (shard < rem ? shard : rem);
  *count = base + (shard < rem ? 1 : 0);
  return 0;
}

}  // extern "C"

namespace {

// librccl is bound at run time (dlopen): the library also has to load where RCCL is not installed, and a group of ONE
// device needs no collective library at all.
struct Rccl {
  void* h = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load(std::string& err) {
    if (h) return true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { err = std::string("librccl.so not loadable: ") + dlerror(); return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
    AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
    GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !AllGather || !GetErrorString) {
      err = "librccl.so lacks an expected symbol";
      dlclose(h);
      h = nullptr;
      return false;
    }
    return true;
  }
};
Rccl g_rccl;
constexpr int kNcclDouble = 8;  // ncclFloat64 (rccl.h)

// final posterior mean of every trajectory of a shard, [D][cnt_max] (padding columns zero): the last record of a fixed
// grid, record NSAVED-1 of an adaptive solve
__global__ void pack_final_kernel(const double* __restrict__ mean, const int* __restrict__ nsaved, long n_save_fixed,
                                  int D, long N, long cnt_max, double* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt_max) return;
  const long slot = i < N ? (nsaved ? (long)nsaved[i] - 1 : n_save_fixed - 1) : 0;
  for (int k = 0; k < D; ++k) out[(size_t)k * cnt_max + i] = i < N ? mean[((size_t)slot * D + k) * N + i] : 0.0;
}

}  // namespace

struct odef_group {
  std::vector<odef_ctx*> ctx;
  std::vector<int64_t> first, count;
  int64_t n_total = 0, cnt_max = 0;
  int D = 0;
  std::vector<void*> comm;         // ncclComm_t per device (empty until the first all-gather of a multi-device group)
  std::vector<double*> send, recv; // per device: [D][cnt_max] and [G][D][cnt_max]
  bool gathered = false;
  std::string err;
};

namespace {
int gfail(odef_group* g, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (g) g->err = buf;
  else g_create_error = buf;
  return -1;
}
template <class F>
int for_all_then_complete(odef_group* g, const char* what, F&& launch) {
  for (odef_ctx* c : g->ctx) c->defer = true;
  int rc = 0;
  size_t launched = 0;
  for (; launched < g->ctx.size() && rc == 0; ++launched) {
    rc = launch(g->ctx[launched], (int)launched);
    if (rc) gfail(g, "%s, shard %zu: %s", what, launched, g->ctx[launched]->err.c_str());
  }
  for (size_t k = 0; k < g->ctx.size(); ++k) {
    odef_ctx* c = g->ctx[k];
    c->defer = false;
    if (c->pending && set_device(c) == 0 && complete_pending(c) != 0 && rc == 0)
      rc = gfail(g, "%s, shard %zu: %s", what, k, c->err.c_str());
  }
  g->gathered = false;
  return rc;
}
}  // namespace

extern "C" {

const char* odef_group_last_error(const odef_group* g) { return g ? g->err.c_str() : g_create_error.c_str(); }

int odef_group_layout(int64_t n_traj, int32_t n_devices, int32_t state_dim, int64_t* first, int64_t* count, int64_t* cnt_max,
                      int64_t* block_doubles) {
  if (n_devices < 1 || n_traj < n_devices || state_dim < 1) return -1;
  int64_t longest = 0;
  for (int32_t k = 0; k < n_devices; ++k) {
    int64_t f = 0, c = 0;
    if (odef_shard_range(n_traj, n_devices, k, &f, &c) != 0) return -1;
    if (first) first[k] = f;
    if (count) count[k] = c;
    longest = c > longest ? c : longest;
  }
  if (cnt_max) *cnt_max = longest;
  if (block_doubles) *block_doubles = (int64_t)state_dim * longest;
  return 0;
}

int odef_unpad_gathered(const double* gathered, int32_t n_devices, int32_t state_dim, int64_t n_traj, double* dst) {
  if (!gathered || !dst) return -1;
  int64_t cnt_max = 0, blk = 0;
  if (odef_group_layout(n_traj, n_devices, state_dim, nullptr, nullptr, &cnt_max, &blk) != 0) return -1;
  for (int32_t k = 0; k < n_devices; ++k) {  // drop the padding columns of the shorter shards
    int64_t f = 0, c = 0;
    odef_shard_range(n_traj, n_devices, k, &f, &c);
    for (int32_t r = 0; r < state_dim; ++r)
      std::memcpy(dst + (size_t)r * (size_t)n_traj + (size_t)f, gathered + (size_t)k * (size_t)blk + (size_t)r * (size_t)cnt_max,
                  (size_t)c * sizeof(double));
  }
  return 0;
}

int odef_group_create(odef_group** out, const odef_config* cfg, int32_t n_devices, const int32_t* device_ids) {
  if (!out || !cfg) return gfail(nullptr, "odef_group_create: null argument");
  *out = nullptr;
  if (n_devices < 1) return gfail(nullptr, "odef_group_create: n_devices must be >= 1");
  if (cfg->n_traj < n_devices) return gfail(nullptr, "odef_group_create: fewer trajectories (%lld) than devices (%d)", (long long)cfg->n_traj, n_devices);
  odef_group* g = new odef_group();
  g->n_total = cfg->n_traj;
  for (int k = 0; k < n_devices; ++k) {
    int64_t first = 0, count = 0;
    odef_shard_range(cfg->n_traj, n_devices, k, &first, &count);
    odef_config sc = *cfg;
    sc.n_traj = count;
    sc.device = device_ids ? device_ids[k] : k;
    odef_ctx* c = nullptr;
    if (odef_create(&c, &sc) != 0) {
      gfail(nullptr, "odef_group_create: shard %d on device %d: %s", k, sc.device, g_create_error.c_str());
      odef_group_destroy(g);
      return -1;
    }
    g->ctx.push_back(c);
    g->first.push_back(first);
    g->count.push_back(count);
    if (count > g->cnt_max) g->cnt_max = count;
    g->D = c->D;
  }
  *out = g;
  return 0;
}

void odef_group_destroy(odef_group* g) {
  if (!g) return;
  for (size_t k = 0; k < g->ctx.size(); ++k) {
    (void)hipSetDevice(g->ctx[k]->device);
    if (k < g->comm.size() && g->comm[k] && g_rccl.CommDestroy) g_rccl.CommDestroy(g->comm[k]);
    if (k < g->send.size() && g->send[k]) (void)hipFree(g->send[k]);
    if (k < g->recv.size() && g->recv[k]) (void)hipFree(g->recv[k]);
    odef_destroy(g->ctx[k]);
  }
  delete g;
}

int32_t odef_group_size(const odef_group* g) { return g ? (int32_t)g->ctx.size() : -1; }
odef_ctx* odef_group_ctx(odef_group* g, int32_t shard) {
  return (g && shard >= 0 && shard < (int32_t)g->ctx.size()) ? g->ctx[shard] : nullptr;
}
int odef_group_shard(const odef_group* g, int32_t shard, int64_t* first, int64_t* count) {
  if (!g || shard < 0 || shard >= (int32_t)g->ctx.size() || !first || !count) return -1;
  *first = g->first[shard];
  *count = g->count[shard];
  return 0;
}

int odef_group_set_problem(odef_group* g, const double* u0, const double* p, double t0) {
  if (!g || !u0) return gfail(g, "odef_group_set_problem: null argument");
  for (size_t k = 0; k < g->ctx.size(); ++k) {
    odef_ctx* c = g->ctx[k];
    const double* pk = (p && !c->cfg.params_shared) ? p + (size_t)g->first[k] * c->np : p;
    if (odef_set_problem(c, u0 + (size_t)g->first[k] * c->d, pk, t0) != 0)
      return gfail(g, "odef_group_set_problem, shard %zu: %s", k, c->err.c_str());
  }
  return 0;
}

int odef_group_set_problem_perturbed(odef_group* g, const double* base_u0, const double* p, double t0, double scale,
                                     uint64_t seed, int32_t n_perturbed) {
  if (!g) return -1;
  for (size_t k = 0; k < g->ctx.size(); ++k)  // global numbering: the ensemble does not depend on the number of shards
    if (odef_set_problem_perturbed(g->ctx[k], base_u0, p, t0, scale, seed, g->first[k], n_perturbed) != 0)
      return gfail(g, "odef_group_set_problem_perturbed, shard %zu: %s", k, g->ctx[k]->err.c_str());
  return 0;
}

int odef_group_solve_fixed(odef_group* g, const double* tgrid, int64_t n_t) {
  if (!g) return -1;
  return for_all_then_complete(g, "odef_group_solve_fixed", [&](odef_ctx* c, int) { return odef_solve_fixed(c, tgrid, n_t); });
}
int odef_group_solve_adaptive(odef_group* g, double t1, double abstol, double reltol, double dt0, const odef_controller* ctrl,
                              int64_t max_steps) {
  if (!g) return -1;
  return for_all_then_complete(g, "odef_group_solve_adaptive",
                               [&](odef_ctx* c, int) { return odef_solve_adaptive(c, t1, abstol, reltol, dt0, ctrl, max_steps); });
}
int odef_group_smooth(odef_group* g) {
  if (!g) return -1;
  return for_all_then_complete(g, "odef_group_smooth", [&](odef_ctx* c, int) { return odef_smooth(c); });
}

int odef_allgather(odef_group* g, int smoothed) {
  if (!g) return -1;
  const int G = (int)g->ctx.size();
  const size_t blk = (size_t)g->D * (size_t)g->cnt_max;  // doubles per shard block
  if (g->send.empty()) {
    g->send.assign(G, nullptr);
    g->recv.assign(G, nullptr);
    for (int k = 0; k < G; ++k) {
      odef_ctx* c = g->ctx[k];
      if (set_device(c)) return gfail(g, "odef_allgather: %s", c->err.c_str());
      if (hipMalloc((void**)&g->send[k], blk * sizeof(double)) != hipSuccess ||
          hipMalloc((void**)&g->recv[k], blk * G * sizeof(double)) != hipSuccess)
        return gfail(g, "odef_allgather: out of device memory on device %d", c->device);
    }
  }
  // shards that share a device (a test rig with fewer GPUs than shards) cannot form an RCCL communicator: the gather
  // is then device-to-device copies
  bool distinct = true;
  for (int a = 0; a < G; ++a)
    for (int b = a + 1; b < G; ++b) distinct = distinct && g->ctx[a]->device != g->ctx[b]->device;
  if (g->comm.empty() && distinct) {
    std::string lerr;
    if (g_rccl.load(lerr)) {
      std::vector<int> devs(G);
      for (int k = 0; k < G; ++k) devs[k] = g->ctx[k]->device;
      g->comm.assign(G, nullptr);
      const int rc = g_rccl.CommInitAll(g->comm.data(), G, devs.data());
      if (rc != 0) {
        g->comm.clear();
        return gfail(g, "odef_allgather: ncclCommInitAll over %d devices failed: %s", G, g_rccl.GetErrorString(rc));
      }
    } else if (G > 1) {
      return gfail(g, "odef_allgather: %s", lerr.c_str());
    }
  }
  // pack the final means of every shard
  for (int k = 0; k < G; ++k) {
    odef_ctx* c = g->ctx[k];
    if (!c->solved) return gfail(g, "odef_allgather: shard %d has not been solved", k);
    const int f = smoothed ? ODEF_F_SMOOTH_MEAN : ODEF_F_MEAN;
    if (smoothed && !c->smoothed_done) return gfail(g, "odef_allgather: smoothed means requested but odef_group_smooth has not run");
    if (set_device(c)) return gfail(g, "odef_allgather: %s", c->err.c_str());
    const long N = (long)c->cfg.n_traj;
    hipLaunchKernelGGL(pack_final_kernel, dim3((unsigned)((g->cnt_max + 255) / 256)), dim3(256), 0, c->stream,
                       (const double*)c->f[f].ptr, c->adaptive ? (const int*)c->f[ODEF_F_NSAVED].ptr : (const int*)nullptr,
                       c->n_save, c->D, N, (long)g->cnt_max, g->send[k]);
  }
  // the one collective of the path: an all-gather over xGMI (RCCL), every device ends with all shards' blocks
  if (!g->comm.empty()) {
    int rc = g_rccl.GroupStart();
    for (int k = 0; k < G && rc == 0; ++k) {
      (void)hipSetDevice(g->ctx[k]->device);
      rc = g_rccl.AllGather(g->send[k], g->recv[k], blk, kNcclDouble, g->comm[k], g->ctx[k]->stream);
    }
    const int rc2 = g_rccl.GroupEnd();
    if (rc == 0) rc = rc2;
    if (rc != 0) return gfail(g, "odef_allgather: ncclAllGather failed: %s", g_rccl.GetErrorString(rc));
  } else {  // one device without RCCL, or shards sharing a device: the gather is a set of copies
    for (int k = 0; k < G; ++k)
      if (hipStreamSynchronize(g->ctx[k]->stream) != hipSuccess) return gfail(g, "odef_allgather: synchronisation failed");
    for (int k = 0; k < G; ++k)
      for (int j = 0; j < G; ++j)
        if (hipMemcpyAsync(g->recv[k] + (size_t)j * blk, g->send[j], blk * sizeof(double), hipMemcpyDeviceToDevice,
                           g->ctx[k]->stream) != hipSuccess)
          return gfail(g, "odef_allgather: device copy failed");
  }
  for (int k = 0; k < G; ++k) {
    (void)hipSetDevice(g->ctx[k]->device);
    if (hipStreamSynchronize(g->ctx[k]->stream) != hipSuccess) return gfail(g, "odef_allgather: synchronisation of device %d failed", g->ctx[k]->device);
  }
  g->gathered = true;
  return 0;
}

int odef_group_get_gathered(odef_group* g, int32_t shard, double* host_dst /* [D][n_total] */, void** dev_ptr, size_t* dev_bytes) {
  if (!g || shard < 0 || shard >= (int32_t)g->ctx.size()) return -1;
  if (!g->gathered) return gfail(g, "odef_group_get_gathered: call odef_allgather first");
  const int G = (int)g->ctx.size();
  const size_t blk = (size_t)g->D * (size_t)g->cnt_max;
  if (dev_ptr) *dev_ptr = g->recv[shard];
  if (dev_bytes) *dev_bytes = blk * G * sizeof(double);
  if (host_dst) {
    odef_ctx* c = g->ctx[shard];
    if (set_device(c)) return gfail(g, "odef_group_get_gathered: %s", c->err.c_str());
    std::vector<double> tmp(blk * G);
    if (hipMemcpy(tmp.data(), g->recv[shard], tmp.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      return gfail(g, "odef_group_get_gathered: copy from device %d failed", c->device);
    if (odef_unpad_gathered(tmp.data(), G, g->D, g->n_total, host_dst) != 0) return gfail(g, "odef_group_get_gathered: inconsistent layout");
  }
  return 0;
}

}  // extern "C"

namespace odef {
namespace {
thread_local char g_last_kernel[192] = "";
}
void note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_last_kernel, sizeof g_last_kernel, fmt, ap);
  va_end(ap);
}
const char* last_kernel() { return g_last_kernel; }

int launch_filter(int rhs, int q, int ek1, int adaptive, const FilterParams& P, hipStream_t s) {
  switch (rhs) {
    case ODEF_RHS_FHN: return launch_filter_fhn(q, ek1, adaptive, P, s);
    case ODEF_RHS_LORENZ63: return launch_filter_lorenz63(q, ek1, adaptive, P, s);
    case ODEF_RHS_LOTKA_VOLTERRA: return launch_filter_lotka_volterra(q, ek1, adaptive, P, s);
    case ODEF_RHS_VANDERPOL: return launch_filter_vanderpol(q, ek1, adaptive, P, s);
    case ODEF_RHS_LINEAR: return launch_filter_linear(q, ek1, adaptive, P, s);
    default: return -2;
  }
}
const TeamLaunch* team_launch(int rhs_id) {
  return rhs_id == ODEF_RHS_PLEIADES ? team_pleiades() : rhs_id == ODEF_RHS_LORENZ96 ? team_lorenz96() : nullptr;
}
int launch_smooth(int d, int q, const SmoothParams& P, hipStream_t s) {
  if (d == 2) return launch_smooth_d2(q, P, s);
  if (d == 3) return launch_smooth_d3(q, P, s);
  return -2;
}
}  // namespace odef
