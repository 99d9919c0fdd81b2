// Per-trajectory drivers: fixed-step filter, adaptive filter, RTS smoother.
// One call = the whole time loop of one trajectory (one GPU lane).  The loop that the
// reference leaves to OrdinaryDiffEq.solve! (loopheader! -> perform_step! -> loopfooter! ->
// savevalues!) runs on the device.
#pragma once
#include "ek_math.h"
#include "rhs.h"

namespace odef {

struct Controller {  // mirrors odef_controller
  double beta1, beta2, gamma, qmin, qmax, qsteady_min, qsteady_max, qoldinit, dtmin, dtmax;
};

struct FilterParams {
  PriorConsts pc;
  // problem
  const double* u0;  // [d][N]
  const double* p;   // [np][N] or [np]
  int p_shared;
  long N;
  // fixed grid: preconditioner tables (one per distinct h, see precond_fill) and, per step, which one
  const double* ptab;    // [n_tables][kTabStride]
  const int* tab_idx;    // [nsteps]
  const double* hs;      // [nsteps] step sizes
  long nsteps;
  // adaptive
  double t0, t1, abstol, reltol, dt0;
  Controller ctrl;
  long max_save;  // capacity of the save axis (adaptive)
  // options
  int everystep, fixed_diffusion, want_loglik;
  int stagger;  // start skew between wavefronts in units of s_sleep (64 clocks); 0 = off
  // outputs
  double* mean;    // [n_save][D][N]
  double* cov;     // [n_save][TRI][N]
  double* diff;    // [n_save][N]
  double* tsave;   // [n_save][N] (adaptive only)
  double* loglik;  // [N]
  int* naccept;
  int* nreject;
  int* nf;
  int* njac;
  int* nsaved;
  int* retcode;
  // workgroup-per-trajectory matrix-core filter only (filter_mfma.h), every step saved: covariance records written
  // trajectory-major into this stage (record_stage.h) and moved to `cov` by the host afterwards; null: written in place
  double* cov_stage;  // [n_save][N][stage_ld]
  long stage_ld;
};

// Row store: `base` is a wave-uniform pointer to element [field row 0][first trajectory of the
// wavefront]; consecutive field rows are `N` doubles apart; every lane writes its own column.
// On the GPU this is a buffer store with the row offset in an SGPR (soffset) and lane*8 in
// voffset: no per-store VALU address arithmetic, 512 contiguous bytes per wave-instruction.
#ifdef ODEF_HOST_EMUL
struct RowStore {
  double* p;
  size_t n;
  RowStore(double* base, size_t N, size_t /*rows*/, unsigned lane) : p(base + lane), n(N) {}
  void put(double v) { *p = v; p += n; }
  RowStore at_row(int row) const {  // (of a store that has not been advanced by put)
    RowStore r = *this;
    r.p = p + (size_t)row * n;
    return r;
  }
};
#else
// Cache-policy bits of the record stores (gfx940+: bit 0 = sc0, bit 1 = nt, bit 4 = sc1).  The records are a pure
// output stream, so they are stored non-temporal; measured on the headline kernel: nt 8.98 ms, default 9.10,
// sc0 9.49, sc1 (any combination) 10.2-10.5.
#ifndef ODEF_STORE_AUX
#define ODEF_STORE_AUX 2
#endif
struct RowStore {
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff, soff, step;
  __device__ RowStore(double* base, size_t N, size_t rows, unsigned lane)
      : rs(__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(rows * N * sizeof(double)), 0x00020000)),
        voff(lane * 8u), soff(0u), step((unsigned)(N * sizeof(double))) {}
  __device__ void put(double v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, soff, ODEF_STORE_AUX);
    soff += step;
#ifndef ODEF_ROWSTORE_FREE_OFFSET
    asm volatile("" : "+s"(soff));  // keep the row offset a running scalar (one s_add per store) instead of
                                    // dozens of hoisted loop-invariant offsets that would spill the SGPR file
                                    // (ODEF_ROWSTORE_FREE_OFFSET: run-time compiled kernels of large state dimension,
                                    // where pinning it to an SGPR makes the register allocator give up)
#endif
  }
  // a store positioned at field row `row` of the same record (one scalar multiply; put() then walks on from there)
  __device__ RowStore at_row(int row) const {
    RowStore r = *this;
    r.soff = (unsigned)row * step;
    return r;
  }
};
#endif

// Row load, the mirror image of RowStore: consecutive rows of one record field, every lane its own column; the row
// offset is a running SGPR, so a record is read with no per-access VALU address arithmetic.
#ifdef ODEF_HOST_EMUL
struct RowLoad {
  const double* p;
  size_t n;
  RowLoad(const double* base, size_t N, size_t /*rows*/, unsigned lane) : p(base + lane), n(N) {}
  double get() {
    const double v = *p;
    p += n;
    return v;
  }
};
#else
struct RowLoad {
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff, soff, step;
  __device__ RowLoad(const double* base, size_t N, size_t rows, unsigned lane)
      : rs(__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(rows * N * sizeof(double)), 0x00020000)),
        voff(lane * 8u), soff(0u), step((unsigned)(N * sizeof(double))) {}
  __device__ double get() {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    soff += step;
    asm volatile("" : "+s"(soff));
    return __builtin_bit_cast(double, r);
  }
};
#endif

// Stores one saved record of one trajectory.  `i0` (first trajectory of the wavefront) and
// `slot` are wave-uniform, `lane` is the lane index.
template <int D, int TRI>
__device__ inline void store_state(const FilterParams& P, long slot, long i0, unsigned lane, const double (&m)[D],
                                   const double (&C)[TRI], double diffusion) {
  const size_t N = (size_t)P.N;
  RowStore sm(P.mean + ((size_t)slot * D * N + i0), N, D, lane);
#pragma unroll
  for (int k = 0; k < D; ++k) sm.put(m[k]);
  RowStore sc(P.cov + ((size_t)slot * TRI * N + i0), N, TRI, lane);
#pragma unroll
  for (int k = 0; k < TRI; ++k) sc.put(C[k]);
  RowStore sd(P.diff + ((size_t)slot * N + i0), N, 1, lane);
  sd.put(diffusion);
}

// Sink of EKStep::run that stores each value of the step's record as soon as it exists.
struct RecordSink {
  RowStore sm, sc;
  __device__ inline void mean(double v) { sm.put(v); }
  __device__ inline void cov(double v) { sc.put(v); }
  __device__ inline void tick() {}
};

// Sink that stores the record of the PREVIOUS step -- the inputs (m, C) of the running step, which stay in registers
// until their own store has been issued -- one value per tick(), i.e. spread evenly over the arithmetic of the step.
// A wavefront can have 63 vector-memory operations in flight and a record is 91 stores at D = 12, so a burst at the
// end of a step parks the wave until 28 of them have completed.  For a LARGE ensemble that wait is hidden behind the
// HBM-bound stream anyway (same time either way, and the lagged form costs 10 % more VALU instructions in AGPR
// moves); for a SMALL ensemble (one wave on a few SIMDs, memory system idle) it is exposed latency: at 4 096
// trajectories 6.2 ms with the burst, 5.8 ms lagged (4.5 ms without stores at all; the rest is the instructions of
// the stores themselves).  `next` is a compile-time constant at every call site once run() is unrolled.
template <int D, int TRI>
struct LaggedSink {
  const double (&m)[D];
  const double (&C)[TRI];
  double diffusion;
  RowStore sm, sc, sd;
  int next;
  __device__ inline void mean(double) {}
  __device__ inline void cov(double) {}
  __device__ inline void tick() {
    if (next < D) sm.put(m[next]);
    else if (next < D + TRI) sc.put(C[next - D]);
    else if (next == D + TRI) sd.put(diffusion);
    ++next;
  }
  __device__ inline void flush() {
#pragma unroll
    for (int k = 0; k <= D + TRI; ++k)
      if (k >= next) tick();
  }
};

template <int D>
__device__ inline bool all_finite(const double (&m)[D]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < D; ++k) ok = ok && (fabs(m[k]) <= 1.79769313486231570815e+308);
  return ok;
}

// EVERY: every step is saved (compile-time: with a run-time flag each of the 91 stores of a step sat behind
// its own branch, which cost 10 % in instructions and scheduling).
// LAG (with EVERY): the record of step n is stored while step n + 1 runs (LaggedSink) -- for small ensembles.
template <class RHS, int q, bool IS_EK1, bool EVERY, bool LAG = false>
__device__ inline void filter_fixed_lane(const FilterParams& P, long i0, unsigned lane) {
  const long i = i0 + lane;
  using S = EKStep<RHS, q, IS_EK1>;
  constexpr int d = S::d, D = S::D, TRI = S::TRI, np = RHS::np;
  double pl[np > 0 ? np : 1];
#pragma unroll
  for (int k = 0; k < np; ++k) pl[k] = P.p_shared ? P.p[k] : P.p[(size_t)k * P.N + i];
  double u0[d];
#pragma unroll
  for (int a = 0; a < d; ++a) u0[a] = P.u0[(size_t)a * P.N + i];

  double m[D], C[TRI];
  taylor_init<RHS, q>(u0, pl, m);
#pragma unroll
  for (int k = 0; k < TRI; ++k) C[k] = 0.0;
  if constexpr (EVERY && !LAG) store_state<D, TRI>(P, 0, i0, lane, m, C, 0.0);

  double loglik = 0.0, gdiff = 0.0;
  LogDetAcc lda;  // log det S of the steps, multiplied up (ek_math.h): det S = (prod |R_kk|)^2
  lda.init();
  int chol_fix = 0;
  for (long n = 0; n < P.nsteps; ++n) {
    const GlobalTab tab{P.ptab + (size_t)uniform_load(P.tab_idx + n) * kTabStride};  // wave-uniform, scalar loads
    double m2[D], C2[TRI], es[d];
    StepAux aux;
    aux.chol_fix = 0;
    const size_t Nn = (size_t)P.N;
    if constexpr (EVERY && LAG) {  // record n = the inputs of this step
      LaggedSink<D, TRI> sink{m, C, gdiff,
                              RowStore(P.mean + ((size_t)n * D * Nn + i0), Nn, D, lane),
                              RowStore(P.cov + ((size_t)n * TRI * Nn + i0), Nn, TRI, lane),
                              RowStore(P.diff + ((size_t)n * Nn + i0), Nn, 1, lane), 0};
      S::run(P.pc, pl, tab, P.fixed_diffusion, P.want_loglik != 0, (int)n, gdiff, m, C, m2, C2, es, aux, sink);
      sink.flush();
    } else if constexpr (EVERY) {
      RecordSink sink{RowStore(P.mean + ((size_t)(n + 1) * D * Nn + i0), Nn, D, lane),
                      RowStore(P.cov + ((size_t)(n + 1) * TRI * Nn + i0), Nn, TRI, lane)};
      S::run(P.pc, pl, tab, P.fixed_diffusion, P.want_loglik != 0, (int)n, gdiff, m, C, m2, C2, es, aux, sink);
    } else {
      NoSink nosink;
      S::run(P.pc, pl, tab, P.fixed_diffusion, P.want_loglik != 0, (int)n, gdiff, m, C, m2, C2, es, aux, nosink);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) m[k] = m2[k];
#pragma unroll
    for (int k = 0; k < TRI; ++k) C[k] = C2[k];
    loglik += aux.loglik;
    if (P.want_loglik) lda.mul(aux.det);
    gdiff = aux.sigma2_global;
    chol_fix += aux.chol_fix;
    if constexpr (EVERY && !LAG) {
      RowStore sd(P.diff + ((size_t)(n + 1) * Nn + i0), Nn, 1, lane);
      sd.put(gdiff);
    }
  }
  if constexpr (!EVERY) store_state<D, TRI>(P, 0, i0, lane, m, C, gdiff);
  if constexpr (EVERY && LAG) store_state<D, TRI>(P, P.nsteps, i0, lane, m, C, gdiff);  // the last record
  P.loglik[i] = P.want_loglik ? loglik - lda.log_value() : loglik;  // -1/2 log det S = -log prod |R_kk|
  P.naccept[i] = (int)P.nsteps;
  P.nreject[i] = 0;
  P.nf[i] = (int)P.nsteps;
  P.njac[i] = IS_EK1 ? (int)P.nsteps : 0;
  P.nsaved[i] = EVERY ? (int)P.nsteps + 1 : 1;
  (void)chol_fix;
  P.retcode[i] = all_finite<D>(m) ? 0 /*Success*/ : 3 /*Unstable*/;
}

// h^(-q-1/2) without libm pow (adaptive steps differ per lane, no host table possible)
template <int q>
__device__ inline double precond_val(double h) {
  double hq = 1.0;
#pragma unroll
  for (int k = 0; k < q; ++k) hq *= h;
  return 1.0 / (hq * sqrt(h));
}

// Per-lane (non-uniform slot) record store / reload for the adaptive filter: plain global accesses with
// per-lane addresses (a buffer descriptor built from a lane-varying slot would be waterfalled per store).
template <int D, int TRI>
__device__ inline void store_state_scatter(const FilterParams& P, long slot, long i, const double (&m)[D],
                                           const double (&C)[TRI], double diffusion, double t) {
  const size_t N = (size_t)P.N;
  double* pm = P.mean + ((size_t)slot * D * N + i);
#pragma unroll
  for (int k = 0; k < D; ++k) { *pm = m[k]; pm += N; }
  double* pcv = P.cov + ((size_t)slot * TRI * N + i);
#pragma unroll
  for (int k = 0; k < TRI; ++k) { *pcv = C[k]; pcv += N; }
  P.diff[(size_t)slot * N + i] = diffusion;
  P.tsave[(size_t)slot * N + i] = t;
}
template <int D, int TRI>
__device__ inline void load_state_scatter(const FilterParams& P, long slot, long i, double (&m)[D], double (&C)[TRI]) {
  const size_t N = (size_t)P.N;
  const double* pm = P.mean + ((size_t)slot * D * N + i);
#pragma unroll
  for (int k = 0; k < D; ++k) { m[k] = *pm; pm += N; }
  const double* pcv = P.cov + ((size_t)slot * TRI * N + i);
#pragma unroll
  for (int k = 0; k < TRI; ++k) { C[k] = *pcv; pcv += N; }
}

// Adaptive filter: perform_step! + error estimate (src/perform_step.jl:78-92) + the PI
// controller of OrdinaryDiffEq (third-party; exponents src/alg_utils.jl:23-24).
// Only ONE copy of the state is kept in registers: the candidate x_filt overwrites it, and the rare
// rejected step re-reads the previous record.
// Record layout: ONE RECORD PER ATTEMPTED STEP.  All lanes of a wavefront attempt their k-th step in the same
// loop iteration, so record k of every lane is written by the same store instructions: full 512-byte rows.  A
// rejected attempt writes the unchanged state again with the unchanged time -- a zero-length step, which the
// smoother, the dense output and the sampler skip exactly as the reference skips duplicated save times
// (src/smoothing.jl:13-16); the host mirror drops those records when it builds sol.t / sol.u.  Storing accepted
// steps only (slot = accepted count, different per lane after the first rejection) splits every store
// instruction over 3-4 rows and ran 3x slower (1.1 TB/s of scattered 8-byte writes).
template <class RHS, int q, bool IS_EK1>
__device__ inline void filter_adaptive_lane(const FilterParams& P, long i0, unsigned lane) {
  const long i = i0 + lane;
  using S = EKStep<RHS, q, IS_EK1>;
  constexpr int d = S::d, D = S::D, TRI = S::TRI, np = RHS::np, NB = q + 1;
  double pl[np > 0 ? np : 1];
#pragma unroll
  for (int k = 0; k < np; ++k) pl[k] = P.p_shared ? P.p[k] : P.p[(size_t)k * P.N + i];
  double u0[d];
#pragma unroll
  for (int a = 0; a < d; ++a) u0[a] = P.u0[(size_t)a * P.N + i];

  double m[D], C[TRI];
  taylor_init<RHS, q>(u0, pl, m);
#pragma unroll
  for (int k = 0; k < TRI; ++k) C[k] = 0.0;
  store_state_scatter<D, TRI>(P, 0, i, m, C, 0.0, P.t0);

  double ucur[d];
#pragma unroll
  for (int a = 0; a < d; ++a) ucur[a] = u0[a];
  const Controller& ct = P.ctrl;
  double t = P.t0, h = P.dt0, qold = ct.qoldinit, q11 = 1.0, log_qold = log(ct.qoldinit), log_eest = 0.0;
  double loglik = 0.0, gdiff = 0.0;
  LogDetAcc lda;
  lda.init();
  int naccept = 0, nreject = 0, nsaved = 1, ret = 0;
  const long max_attempts = 20 * P.max_save + 1000;
  long attempts = 0;
  while (t < P.t1) {
    if (nsaved >= P.max_save || attempts >= max_attempts) { ret = 1; break; }  // MaxIters
    ++attempts;
    h = fmin(h, ct.dtmax);
    h = fmin(h, P.t1 - t);  // tstop clipping
    if (!(h > ct.dtmin)) { ret = 2; break; }  // DtLessThanMin
    double tabv[kTabStride];
    precond_table_fast<q, NB, true>(h, tabv);  // no division, no libm pow: rebuilt at every attempted step
    const LocalTab tab{tabv};
    double es[d];
    StepAux aux;
    aux.chol_fix = 0;
    {
      double m2[D], C2[TRI];
      NoSink nosink;
      S::run(P.pc, pl, tab, P.fixed_diffusion, P.want_loglik != 0, naccept, gdiff, m, C, m2, C2, es, aux, nosink);
#pragma unroll
      for (int k = 0; k < D; ++k) m[k] = m2[k];
#pragma unroll
      for (int k = 0; k < TRI; ++k) C[k] = C2[k];
    }
    // DiffEqBase.calculate_residuals! + ODE_DEFAULT_NORM (src/perform_step.jl:78-84)
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < d; ++r) {
      const double e = h * es[r] * rcp_pos(P.abstol + fmax(fabs(ucur[r]), fabs(m[r])) * P.reltol);
      acc += e * e;
    }
    double EEst = sqrt(acc * (1.0 / d));
    if (!(EEst == EEst) || !(fabs(EEst) <= 1.79769313486231570815e+308)) EEst = INFINITY;
#pragma unroll
    for (int r = 0; r < d; ++r) ucur[r] = m[r];  // integ.u .= u_filt (src/perform_step.jl:86), also when rejected
    // stepsize_controller! (PI)
    double qq;
    if (EEst == 0.0) {
      qq = 1.0 / ct.qmax;
    } else {
      // x^b = exp(b log x) for the positive arguments of the controller (1e-15 relative, a third of pow()'s
      // instructions); log(qold) is carried from the attempt that set qold, q11 alone is needed only after a rejection
      log_eest = log(EEst);
      qq = exp(ct.beta1 * log_eest - ct.beta2 * log_qold);
      qq = fmax(1.0 / ct.qmax, fmin(1.0 / ct.qmin, qq / ct.gamma));
    }
    const bool accepted = EEst <= 1.0;  // OrdinaryDiffEq accepts on <=
    if (!(EEst < 1.0)) {
      // x_filt is not committed (src/perform_step.jl:89): cache.x stays P^-1 (P x) of the old state (:73)
      load_state_scatter<D, TRI>(P, nsaved - 1, i, m, C);  // the previous record holds the current accepted state
#pragma unroll
      for (int k = 0; k < D; ++k) m[k] = tab[kTabPIJ + k / d] * (tab[kTabPJ + k / d] * m[k]);
    }
    if (accepted) {
      if (EEst < 1.0) {
        loglik += aux.loglik;
        if (P.want_loglik) lda.mul(aux.det);
      }
      if (qq <= ct.qsteady_max && qq >= ct.qsteady_min) qq = 1.0;
      qold = fmax(EEst, ct.qoldinit);
      log_qold = (EEst > ct.qoldinit) ? log_eest : log(ct.qoldinit);
      double tn = t + h;
      if (fabs(tn - P.t1) < 100.0 * 2.220446049250313e-16 * fmax(fabs(tn), fabs(P.t1))) tn = P.t1;
      t = tn;
      gdiff = aux.sigma2_global;
      ++naccept;
      h = h / qq;
    } else {
      ++nreject;
      q11 = (EEst == 0.0) ? 1.0 : exp(ct.beta1 * log_eest);
      h = h / fmin(1.0 / ct.qmin, q11 / ct.gamma);
    }
    // accepted: the new state at the new time; rejected: the old state again at the old time
    store_state_scatter<D, TRI>(P, nsaved, i, m, C, gdiff, t);
    ++nsaved;
    if (accepted && !all_finite<D>(m)) { ret = 3; break; }
  }
  (void)qold;
  P.loglik[i] = P.want_loglik ? loglik - lda.log_value() : loglik;
  P.naccept[i] = naccept;
  P.nreject[i] = nreject;
  P.nf[i] = naccept + nreject;
  P.njac[i] = IS_EK1 ? naccept + nreject : 0;
  P.nsaved[i] = nsaved;
  P.retcode[i] = ret;
}

// ---------------------------------------------------------------------------------------
// Rauch-Tung-Striebel pass (src/smoothing.jl:4-63, src/filtering.jl:136-154).
// Joseph form on covariance storage:  Sigma^s = (I-GA) Sigma (I-GA)' + G (sigma^2 Q + Sigma^s_+) G'
// which is the Gram matrix of the reference's stacked QR factor (src/smoothing.jl:53-57).
// ---------------------------------------------------------------------------------------
struct SmoothParams {
  PriorConsts pc;
  long N;
  long n_save;          // fixed: number of saves; adaptive: capacity
  int adaptive;
  const double* ptab;   // fixed: preconditioner tables
  const int* tab_idx;   // fixed: [n_save-1]
  const double* hs;     // fixed: [n_save-1]
  const double* tsave;  // adaptive: [n_save][N]
  const int* nsaved;    // [N]
  const double* mean;   // filter results
  const double* cov;
  const double* diff;
  double* smean;        // outputs, same layout
  double* scov;
  int* retcode;
  // workgroup-per-trajectory matrix-core smoother only (smooth_mfma.h): covariance records staged trajectory-major, so
  // that a workgroup reads and writes its records as contiguous lines (null: the records are used where they lie)
  double* stage;   // [records stage_s0 ..][N][stage_ld]: filter covariances in, smoothed covariances out (in place)
  long stage_s0;   // save index of the first staged record
  long stage_ld;   // doubles per staged record (the packed triangle rounded up to whole 128-byte lines)
  long stage_hi;   // save index of the last staged record: this launch smooths the records stage_hi .. stage_s0 that a trajectory
                   // has (of its 1 .. n - 2); a trajectory whose last record n - 1 lies in the stage starts its carried state
                   // from it, one whose last record lies above continues from the workspace, one below has nothing to do yet
  // the same pass as a sequence of kernels per record (the default; ODEF_SMOOTH_SPLIT=0 turns it off; smooth_mfma.h): 0 = one persistent launch per block;
  // 1 = set up / carry over the block's state only; 2 = begin record split_sa (unpack, predict) -- everything that follows
  // (factorisation, sweeps, mean, G M G', the smoothed record) runs on chip in rts_smooth_sweeps_kernel, launched behind it
  int split_mode;
  long split_sc, split_sa;  // split_sa: the record of this pair of launches; split_sc == 1: Y' = A X is formed by the on-chip kernel from the record (no hand-over through the workspace)
};

}  // namespace odef
