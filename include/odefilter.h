/* libodefilter_hip.so -- C ABI of the MI355X (gfx950) Gaussian ODE-filter hot path.
 *
 * Drop-in boundary for the Kalman predict / measure / calibrate / update / smooth path of
 * ProbNumDiffEq.jl v0.1.5 (nathanaelbosch/ODEFilters.jl).  Citations are file:line in the
 * reference tree.  A Julia host (`julia/ODEFilterHIP.jl`) binds these
 * entry points with `ccall`; the Python host mirror (`odefilters.jl_amd/host.py`) binds the
 * same symbols with ctypes.  Plain pointers and sizes only; no callbacks into the host.
 *
 * What each entry point replaces in the reference:
 *   odef_create / odef_destroy      alg_cache(alg::GaussianODEFilter, ...)        src/caches.jl:42-114
 *                                   (A, Q, Precond, Proj, work arrays; EK0/EK1 kwargs src/algorithms.jl:23-51)
 *   odef_set_problem*               ODEProblem(f,u0,tspan,p) x EnsembleProblem prob_func; initialize! +
 *                                   initial_update!                              src/perform_step.jl:2-12,
 *                                                                                src/state_initialization.jl:2-53
 *   odef_solve_fixed                solve(prob, alg; adaptive=false, dt) = loop of perform_step!
 *                                                                                src/perform_step.jl:27-93
 *                                   incl. predict!/update!                       src/filtering.jl:17-48,79-91
 *                                   measure!                                     src/perform_step.jl:95-132
 *                                   estimate_diffusion                           src/diffusions.jl:11-36,71-80
 *                                   savevalues!                                  src/integrator_utils.jl:33-48
 *   odef_solve_adaptive             same + estimate_errors                       src/perform_step.jl:78-84,148-158
 *                                   + PI controller (exponents src/alg_utils.jl:23-24; loop is OrdinaryDiffEq's)
 *   odef_smooth                     postamble! -> smooth_all! -> smooth!         src/integrator_utils.jl:2-30,
 *                                                                                src/smoothing.jl:4-63
 *   odef_dense_output               sol(t), GaussianODEFilterPosterior            src/solution.jl:165-214
 *   odef_sample                     sample_states / sample                        src/solution_sampling.jl:15-62
 *   odef_dense_sample               dense_sample_states / dense_sample            src/solution_sampling.jl:63-75
 *   odef_rhs_compile                the user's f / f.jac closure (compiled, not called) src/perform_step.jl:106,116-121
 *   odef_get / odef_get_device      sol.t, sol.x_filt, sol.x_smooth, sol.diffusions, sol.log_likelihood,
 *                                   sol.destats, sol.retcode                     src/solution.jl:8-24
 *   odef_predict / odef_update /
 *   odef_smooth_step                predict!, update!, smooth (pure functions)   src/filtering.jl:17,79,136
 *   odef_ibm / odef_preconditioner  ibm(d,q), preconditioner(T,d,q)              src/priors.jl:7-59,
 *                                                                                src/preconditioning.jl:1-17
 *   odef_group_* / odef_allgather   nothing in the reference (it has no ensemble and no distributed code):
 *                                   the EnsembleProblem axis sharded over the GPUs of one node by ONE host process, the
 *                                   per-trajectory path above unchanged on every shard, one RCCL all-gather at the end
 *
 * Error convention: every function returns 0 on success, <0 on API / HIP failure with a
 * message in odef_last_error().  Numerical trouble is per trajectory (RETCODE field) and
 * never aborts the batch (the reference throws / asserts: src/numerics_tricks.jl:1-6,
 * src/smoothing.jl:25).
 *
 * Device data layout (all double unless noted), N = n_traj, D = d*(order+1), TRI = D(D+1)/2,
 * trajectory index fastest so that one 64-lane wavefront (one lane per trajectory) stores
 * 512 contiguous bytes per field element:
 *   MEAN        [n_save][D][N]     state ordering derivative-major (src/caches.jl:63-64)
 *   COV_TRIL    [n_save][TRI][N]   packed lower triangle, element (i,j), i>=j at i(i+1)/2+j, of
 *                                  Sigma = L L'  (`SquarerootMatrix.mat`, src/squarerootmatrix.jl:16)
 *   DIFFUSION   [n_save][N]        entry s = global diffusion of the step that produced save s (s>=1); [0]=0
 *   T           [n_save] (fixed)   or [n_save][N] (adaptive)
 *   LOGLIK      [N]                sum of per-accepted-step log-likelihoods (src/perform_step.jl:66,91)
 *   NACCEPT, NREJECT, NF, NJAC     int32 [N]   (destats)
 *   NSAVED      int32 [N]          number of valid saves of a trajectory (adaptive)
 *   RETCODE     int32 [N]          odef_retcode
 *   SMOOTH_MEAN / SMOOTH_COV_TRIL  as MEAN / COV_TRIL after odef_smooth
 * odef_get copies a field to host memory in exactly this layout.
 */
#ifndef ODEFILTER_H
#define ODEFILTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODEF_VERSION 100

typedef struct odef_ctx odef_ctx;

typedef enum { ODEF_EK0 = 0, ODEF_EK1 = 1 } odef_alg;
/* diffusionmodel = :dynamic / :fixed / :fixedMAP (src/caches.jl:89-96, src/diffusions.jl:11-36, 46-68, 71-80) */
typedef enum { ODEF_DIFFUSION_DYNAMIC = 0, ODEF_DIFFUSION_FIXED = 1, ODEF_DIFFUSION_FIXED_MAP = 2 } odef_diffusion;
typedef enum {
  ODEF_RHS_FHN = 0,            /* u' = (c(u1 - u1^3/3 + u2), -(u1 - a - b u2)/c), p = (a,b,c) */
  ODEF_RHS_LORENZ63 = 1,       /* p = (sigma, rho, beta) */
  ODEF_RHS_LOTKA_VOLTERRA = 2, /* u' = (a u1 - b u1 u2, -c u2 + d u1 u2), p = (a,b,c,d) */
  ODEF_RHS_VANDERPOL = 3,      /* u' = (u2, mu((1-u1^2)u2 - u1)), p = (mu) */
  ODEF_RHS_LINEAR = 4,         /* u_i' = p_i u_i, d = 2 */
  ODEF_RHS_PLEIADES = 5,       /* 7-body, d = 28, no parameters; workgroup-per-trajectory kernels (matrix cores) */
  ODEF_RHS_LORENZ96 = 6        /* u_i' = (u_{i+1} - u_{i-2}) u_{i-1} - u_i + F, d = 16, p = (F); workgroup-per-trajectory kernels */
} odef_rhs;
typedef enum { ODEF_SAVE_FINAL = 0, ODEF_SAVE_EVERYSTEP = 1 } odef_save_mode;
typedef enum {
  ODEF_RET_SUCCESS = 0,
  ODEF_RET_MAXITERS = 1,
  ODEF_RET_DT_LESS_THAN_MIN = 2,
  ODEF_RET_UNSTABLE = 3,       /* NaN/Inf in the state */
  ODEF_RET_NEGATIVE_VARIANCE = 4
} odef_retcode;
typedef enum {
  ODEF_F_MEAN = 0,
  ODEF_F_COV_TRIL = 1,
  ODEF_F_DIFFUSION = 2,
  ODEF_F_T = 3,
  ODEF_F_LOGLIK = 4,
  ODEF_F_NACCEPT = 5,
  ODEF_F_NREJECT = 6,
  ODEF_F_NF = 7,
  ODEF_F_NJAC = 8,
  ODEF_F_NSAVED = 9,
  ODEF_F_RETCODE = 10,
  ODEF_F_SMOOTH_MEAN = 11,
  ODEF_F_SMOOTH_COV_TRIL = 12,
  ODEF_F_U0 = 13,              /* [d][N] initial values as held on the device */
  ODEF_F_DENSE_MEAN = 14,      /* [n_q][D][N]   result of odef_dense_output */
  ODEF_F_DENSE_COV_TRIL = 15,  /* [n_q][TRI][N] */
  ODEF_F_SAMPLES = 16,         /* [n_save][D][n_samples][N] result of odef_sample */
  ODEF_F_COUNT_
} odef_field;

/* POD mirror of the reference's keyword structs (src/algorithms.jl:23-28,46-51) plus the
 * ensemble shape.  Zero-initialise, set struct_size = sizeof(odef_config). */
typedef struct {
  int32_t struct_size;
  int32_t alg;           /* odef_alg */
  int32_t order;         /* q, 1..ODEF_MAX_ORDER */
  int32_t diffusion;     /* odef_diffusion */
  int32_t smooth;        /* keep what odef_smooth needs (forces ODEF_SAVE_EVERYSTEP) */
  int32_t rhs_id;        /* odef_rhs */
  int32_t d;             /* must equal the registry's dimension for rhs_id */
  int32_t n_params;      /* must equal the registry's parameter count */
  int32_t params_shared; /* 1: one parameter vector for all trajectories; 0: p[N][n_params] */
  int32_t save_mode;     /* odef_save_mode */
  int32_t device;        /* HIP device ordinal, -1 = current device */
  int32_t want_loglik;   /* accumulate sol.log_likelihood (src/perform_step.jl:66) */
  int64_t n_traj;        /* N */
} odef_config;

/* OrdinaryDiffEq PI step-size controller (third-party defaults; exponents from
 * src/alg_utils.jl:23-24).  Pass NULL to odef_solve_adaptive for these defaults. */
typedef struct {
  double beta1, beta2, gamma, qmin, qmax, qsteady_min, qsteady_max, qoldinit, dtmin, dtmax;
} odef_controller;

#define ODEF_MAX_ORDER 5

int odef_version(void);
const char* odef_last_error(const odef_ctx* ctx); /* ctx may be NULL: last error of odef_create */

/* Run-time compiled vector field.  The reference calls the user's Julia closure f (and f.jac or ForwardDiff) in the
 * middle of every step (src/perform_step.jl:106,116-121); a device kernel cannot call back into the host, so the
 * vector field is handed over as HIP C++ SOURCE: the definition of a struct `name` with the interface of the
 * compiled-in registry (odefilters.jl_amd/csrc/rhs.h), placed inside namespace odef:
 *
 *     struct MyField {
 *       static constexpr int d = 3, np = 3;
 *       template <class T>   // T = double in the step, a truncated Taylor jet in the initialisation
 *       __device__ static void f(const T (&u)[3], const double* p, T (&du)[3]) { ... }
 *       __device__ static void jac(const double (&u)[3], const double* p, double (&J)[3][3]) { ... }  // optional:
 *           // without it EK1 differentiates f in forward mode (the reference's ForwardDiff fallback, :119-121)
 *     };
 *
 * A hipcc child process ($ODEFILTER_HIP_HIPCC, else hipcc on PATH, else /opt/rocm/bin/hipcc) compiles the library's
 * own kernels around it for gfx950 (include_dir = directory holding the csrc headers; NULL: $ODEFILTER_HIP_INCLUDE, else
 * ../csrc next to the loaded library, else the build-time location):
 *   d(q+1) <= 20, d <= 10   one lane per trajectory, and for d(q+1) <= 16 also the 16-lanes-per-trajectory kernels that small
 *                           and sharded ensembles use, chosen by ensemble size exactly as for the compiled-in fields
 *                           (dense output and sampling: d(q+1) <= 12 on lanes, <= 32 on row teams);
 *   above, even d <= 32,    the workgroup-per-trajectory kernels on the matrix cores (filter on fixed grids and adaptive,
 *   d(q+1) <= 176           smoother, dense output, sampling) -- built at odef_create for that order and algorithm as a host +
 *                           device shared object that links against this library (tens of seconds to minutes).
 * Returns 0 and a new rhs id (>= 100) for odef_config.rhs_id; on a compile error returns -1 and odef_last_error(NULL) holds
 * the compiler log (d <= 10: the order-1 lane filter is built at once; above: a probe kernel that evaluates f in double, in
 * forward mode and on Taylor jets).  An odd d above state dimension 20 has no kernel: odef_create says so.  Whatever the
 * compiler rejects comes back as an error with its log. */
int odef_rhs_compile(const char* name, const char* source, int32_t d, int32_t n_params, const char* include_dir,
                     int32_t* rhs_id);

int odef_create(odef_ctx** out, const odef_config* cfg);
void odef_destroy(odef_ctx* ctx);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the ctx's own. */
int odef_set_stream(odef_ctx* ctx, void* hip_stream);

/* Initial values.  Host layouts are trajectory-major: u0[N][d], p[N][n_params] (or
 * p[n_params] when params_shared) -- i.e. Julia's d x N column-major matrices. */
int odef_set_problem(odef_ctx* ctx, const double* u0, const double* p, double t0);
/* Same from device memory already in the device layout u0[d][N], p[n_params][N]. */
int odef_set_problem_device(odef_ctx* ctx, const double* d_u0, const double* d_p, double t0);
/* Synthetic ensemble generated on the device (the EnsembleProblem prob_func of SURVEY 8d):
 * u0_i[k] = base_u0[k] + scale*(2U-1) for k < n_perturbed, U = (splitmix64(seed +
 * n_perturbed*(first_index+i) + k) >> 11) * 2^-53.  p = shared parameter vector. */
int odef_set_problem_perturbed(odef_ctx* ctx, const double* base_u0, const double* p, double t0,
                               double scale, uint64_t seed, int64_t first_index, int32_t n_perturbed);

/* Fixed-step filter over the host time grid tgrid[0..n_t-1] (n_t-1 steps for every trajectory). */
int odef_solve_fixed(odef_ctx* ctx, const double* tgrid, int64_t n_t);
/* Adaptive filter from t0 to t1.  ONE RECORD PER ATTEMPTED STEP (at most max_steps per trajectory, NSAVED = attempts
 * + 1): an accepted step stores the new state at the new time, a rejected attempt stores the unchanged state again at
 * the unchanged time.  All lanes of a wavefront then write record k in the same instructions (full 512-byte rows);
 * odef_smooth / odef_dense_output / odef_sample treat the repeated record as the reference treats a duplicated save
 * time (src/smoothing.jl:13-16).  A binding that wants the reference's sol.t / sol.u (accepted steps only,
 * src/integrator_utils.jl:33-48) drops records whose T equals the previous T (host.py EnsembleSolution._order). */
int odef_solve_adaptive(odef_ctx* ctx, double t1, double abstol, double reltol, double dt0,
                        const odef_controller* ctrl, int64_t max_steps);
/* Rauch-Tung-Striebel pass over the stored filter states. */
int odef_smooth(odef_ctx* ctx);

/* Dense output / saveat (src/solution.jl:165-210): posterior of every trajectory at the n_q host times tq
 * (smoothed != 0: the smoothed posterior, needs odef_smooth first).  Results in ODEF_F_DENSE_MEAN /
 * ODEF_F_DENSE_COV_TRIL.  Times before t0 give NaN records (the reference throws).  Any state dimension the solver accepts:
 * <= 12 one lane per (trajectory, time), <= 32 one team of 16 / 32 lanes, the workgroup-per-trajectory path (168) on the MFMA smoother's algebra. */
int odef_dense_output(odef_ctx* ctx, const double* tq, int64_t n_q, int smoothed);

/* Posterior sampling on the saved grid (src/solution_sampling.jl:24-62): n_samples joint draws of the whole state
 * path per trajectory, x_N ~ N(mu_N, S_N) and backwards x_i ~ smooth(x_filt[i], delta(x_{i+1})).  Needs
 * ODEF_SAVE_EVERYSTEP (the reference asserts a smoothing solve, :16); odef_smooth itself is not required.
 * The N(0,1) stream is counter-based and reproducible from `seed` (splitmix64 + Box-Muller, see
 * oracle/odefilter_oracle.py sample_normal); noise_scale = 1 gives samples, 0 the chain of conditional means.
 * Result in ODEF_F_SAMPLES; sample(sol, n) of the reference is its rows 0..d-1.  Any state dimension the solver accepts
 * (<= 12 one lane per (trajectory, sample), <= 32 one team of 16 / 32 lanes, the workgroup-per-trajectory path on the MFMA algebra). */
int odef_sample(odef_ctx* ctx, int64_t n_samples, uint64_t seed, double noise_scale);

/* Posterior sampling on a dense grid (dense_sample_states / dense_sample, src/solution_sampling.jl:63-75): the FILTER
 * posterior is interpolated at the n_q host times tq (as odef_dense_output with smoothed = 0; the reference uses
 * range(t0, t_end, length = 1000)) and the backward sampler of odef_sample runs over those states, the diffusion
 * of an interval looked up by time (:41).  Overwrites ODEF_F_DENSE_MEAN / ODEF_F_DENSE_COV_TRIL; result in
 * ODEF_F_SAMPLES as [n_q][D][n_samples][N].  tq must be non-decreasing and >= t0.  Any state dimension the solver accepts. */
int odef_dense_sample(odef_ctx* ctx, const double* tq, int64_t n_q, int64_t n_samples, uint64_t seed, double noise_scale);

int64_t odef_n_save(const odef_ctx* ctx); /* leading dimension of MEAN/COV_TRIL/DIFFUSION/T */
int odef_field_bytes(const odef_ctx* ctx, int field, size_t* bytes);
int odef_get(odef_ctx* ctx, int field, void* host_dst, size_t bytes);
int odef_get_device(odef_ctx* ctx, int field, void** dev_ptr, size_t* bytes);
/* Use a caller-owned device buffer for an output field (before odef_solve_*). */
int odef_bind_device(odef_ctx* ctx, int field, void* dev_ptr, size_t bytes);
int odef_synchronize(odef_ctx* ctx);

/* Device time of the last filter (which=0) / smoother (which=1) launch, measured with
 * hipEvents on the launch stream; n_launches = kernels launched by that call. */
int odef_kernel_time_ms(odef_ctx* ctx, int which, float* ms, int* n_launches);
/* Name of the kernel that call launched (the dominant one of a multi-kernel pass), as a profiler prints it, e.g.
 * "odef::ek_filter_fixed_kernel<odef::RhsLorenz63, 3, true, true, false>": which kernel serves a configuration depends on
 * the state dimension and the ensemble size (crossovers in csrc/ek_kernels.h), and a benchmark line should name the kernel
 * that ran, not re-derive those thresholds.  NUL-terminated into buf (truncated to n); "" before the first launch. */
int odef_kernel_name(odef_ctx* ctx, int which, char* buf, size_t n);

/* ---- ensemble sharded over the GPUs of one node, single host process -------------------------------
 * Trajectories are independent, so the ensemble is cut into contiguous blocks, one per device; nothing is exchanged
 * while stepping; odef_allgather is the ONE collective (ncclAllGather over xGMI, librccl bound with dlopen at the first
 * call; `ncclCommInitAll`, no MPI, no second process).  A Julia host drives 8 GPUs through these calls alone. */
typedef struct odef_group odef_group;
/* block [first, first + count) of shard `shard` out of n_shards: the first n_traj % n_shards shards are one longer */
int odef_shard_range(int64_t n_traj, int32_t n_shards, int32_t shard, int64_t* first, int64_t* count);
/* Layout of the gathered block (host arithmetic, no device needed): every device ends odef_allgather with
 * [n_devices][state_dim][cnt_max] doubles -- shard k's final means in block k, its `count[k]` columns first, the columns
 * up to cnt_max = the longest shard's count zero-padded (RCCL's all-gather wants equal blocks).  first / count: arrays of
 * n_devices entries (either may be NULL); block_doubles = state_dim * cnt_max.  What a caller that consumes the block on
 * the device (odef_group_get_gathered's dev_ptr) needs to address it. */
int odef_group_layout(int64_t n_traj, int32_t n_devices, int32_t state_dim, int64_t* first, int64_t* count, int64_t* cnt_max,
                      int64_t* block_doubles);
/* ... and the same block with the padding dropped: gathered [n_devices][state_dim][cnt_max] (host memory) ->
 * dst [state_dim][n_traj], trajectory index fastest as everywhere.  odef_group_get_gathered's host path is this function. */
int odef_unpad_gathered(const double* gathered, int32_t n_devices, int32_t state_dim, int64_t n_traj, double* dst);
/* cfg->n_traj is the size of the WHOLE ensemble; cfg->device is ignored; device_ids == NULL: devices 0..n_devices-1 */
int odef_group_create(odef_group** out, const odef_config* cfg, int32_t n_devices, const int32_t* device_ids);
void odef_group_destroy(odef_group* g);
const char* odef_group_last_error(const odef_group* g); /* g may be NULL: last error of odef_group_create */
int32_t odef_group_size(const odef_group* g);
odef_ctx* odef_group_ctx(odef_group* g, int32_t shard);  /* the shard's context (owned by the group): odef_get etc. */
int odef_group_shard(const odef_group* g, int32_t shard, int64_t* first, int64_t* count);
/* u0[N][d], p[N][n_params] (or shared p) of the whole ensemble in host memory; each shard takes its block */
int odef_group_set_problem(odef_group* g, const double* u0, const double* p, double t0);
/* odef_set_problem_perturbed with GLOBAL trajectory numbering: the ensemble does not depend on the number of devices */
int odef_group_set_problem_perturbed(odef_group* g, const double* base_u0, const double* p, double t0, double scale,
                                     uint64_t seed, int32_t n_perturbed);
/* the shards' kernels are launched one after the other and run concurrently; the call returns when all are done */
int odef_group_solve_fixed(odef_group* g, const double* tgrid, int64_t n_t);
int odef_group_solve_adaptive(odef_group* g, double t1, double abstol, double reltol, double dt0,
                              const odef_controller* ctrl, int64_t max_steps);
int odef_group_smooth(odef_group* g);
/* All-gather of the FINAL posterior mean of every trajectory (smoothed == 0: filter, record NSAVED-1; != 0: smoothed):
 * afterwards every device holds [n_shards][D][count_max] doubles (blocks of the shorter shards zero-padded). */
int odef_allgather(odef_group* g, int smoothed);
/* the gathered result as device `shard` holds it: host_dst (may be NULL) receives [D][n_traj] with the padding
 * removed; dev_ptr / dev_bytes (may be NULL) the raw device block */
int odef_group_get_gathered(odef_group* g, int32_t shard, double* host_dst, void** dev_ptr, size_t* dev_bytes);

/* Constants (host side, for tests and host mirrors). A, Q_L: D x D row-major. */
int odef_ibm(int d, int q, double* A, double* Q_L);
int odef_preconditioner(int d, int q, double h, double* P_diag);

/* Step-level entry points on batched Gaussians held in HOST memory (copied to the device,
 * computed by HIP kernels, copied back).  n instances; mu[n][D]; L[n][D][D] row-major
 * square-root factors (Sigma = L L'); A, Q_L: D x D row-major shared by the batch.
 * D <= ODEF_MAX_STEP_DIM. */
#define ODEF_MAX_STEP_DIM 32
int odef_predict(int D, int64_t n, const double* mu, const double* L, const double* A, const double* Q_L,
                 double* mu_out, double* cov_out /* [n][D][D] full symmetric */);
/* measurement Z = N(z, S) with S = H Sigma_pred H' (R = 0, asserted by the reference at src/filtering.jl:81) */
int odef_update(int D, int o, int64_t n, const double* mu_pred, const double* L_pred, const double* H /* [n][o][D] */,
                const double* z /* [n][o] */, double* mu_out, double* cov_out /* [n][D][D] */);
int odef_smooth_step(int D, int64_t n, const double* mu, const double* L, const double* mu_s, const double* L_s,
                     const double* A, const double* Q_L, double* mu_out, double* cov_out);

#ifdef __cplusplus
}
#endif
#endif /* ODEFILTER_H */
