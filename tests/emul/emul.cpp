// TEST INFRASTRUCTURE: host build of the per-lane device source (ek_math.h / ek_lane.h)
// so that the arithmetic the HIP kernels execute can be checked against the oracle on a
// machine without a GPU (and under the CPU sanitizers).  Not part of the product; the
// library never loads this.
#define ODEF_HOST_EMUL 1
#include "../../odefilters.jl_amd/csrc/dispatch.h"
#include "../../odefilters.jl_amd/csrc/smooth_team.h"
#include "../../odefilters.jl_amd/csrc/smooth_rows.h"
#include "../../odefilters.jl_amd/csrc/smooth_lane.h"
#include "../../odefilters.jl_amd/csrc/dense_lane.h"
#include "../../odefilters.jl_amd/csrc/filter_team.h"
#include "../../odefilters.jl_amd/csrc/filter_tiles.h"
#include "../../odefilters.jl_amd/csrc/sample_lane.h"
#include "../../odefilters.jl_amd/csrc/rows_filter.h"
#include "../../odefilters.jl_amd/csrc/rows_smooth.h"
#include <vector>
#include <cstring>

using namespace odef;

struct EmulArgs {
  int rhs, q, ek1, adaptive;
  long N;
  const double *u0, *p;
  int p_shared;
  const double *At, *Qt, *QLt;  // MAXNB*MAXNB each
  const double *hs, *ptab;
  const int* tab_idx;
  long nsteps;
  double t0, t1, abstol, reltol, dt0;
  const double* ctrl;  // 10
  long max_save;
  int everystep, fixed_diffusion, want_loglik;
  double *mean, *cov, *diff, *tsave, *loglik;
  int *naccept, *nreject, *nf, *njac, *nsaved, *retcode;
  // smoother
  double *smean, *scov;
  long n_save;
};

static void fill(const EmulArgs& a, FilterParams& P) {
  std::memcpy(P.pc.At, a.At, sizeof(P.pc.At));
  std::memcpy(P.pc.Qt, a.Qt, sizeof(P.pc.Qt));
  std::memcpy(P.pc.QLt, a.QLt, sizeof(P.pc.QLt));
  P.u0 = a.u0; P.p = a.p; P.p_shared = a.p_shared; P.N = a.N;
  P.hs = a.hs; P.ptab = a.ptab; P.tab_idx = a.tab_idx; P.nsteps = a.nsteps;
  P.t0 = a.t0; P.t1 = a.t1; P.abstol = a.abstol; P.reltol = a.reltol; P.dt0 = a.dt0;
  std::memcpy(&P.ctrl, a.ctrl, sizeof(Controller));
  P.max_save = a.max_save;
  P.everystep = a.everystep > 0; P.fixed_diffusion = a.fixed_diffusion; P.want_loglik = a.want_loglik;
  // everystep == 2: the lagged record stores of the small-ensemble kernel; 3 / -1: the row-team filter (every step / final)
  P.stagger = a.everystep == 2 ? 7 : (a.everystep == 3 || a.everystep == -1) ? 9 : 0;
  P.mean = a.mean; P.cov = a.cov; P.diff = a.diff; P.tsave = a.tsave; P.loglik = a.loglik;
  P.naccept = a.naccept; P.nreject = a.nreject; P.nf = a.nf; P.njac = a.njac; P.nsaved = a.nsaved;
  P.retcode = a.retcode;
}

struct RunFilter {
  const FilterParams& P;
  int adaptive;
  template <class RHS, int q, bool EK1>
  void operator()() {
    for (long i = 0; i < P.N; ++i) {
      const long i0 = (i / 64) * 64;
      if (P.stagger == 9) {  // row-per-lane team filter (rows_filter.h): one 16-lane team per trajectory
        if constexpr (RHS::d * (q + 1) <= 16) {
          std::vector<double> ws(RowsStep<RHS, q, EK1>::kLdsDoublesAdaptive);
          const RowsTeam tm{i, i, true, 0, ws.data(), nullptr};
          if (adaptive) rows_filter_adaptive<RHS, q, EK1>(P, tm);
          else if (P.everystep) rows_filter_fixed<RHS, q, EK1, true>(P, tm);
          else rows_filter_fixed<RHS, q, EK1, false>(P, tm);
        }
        continue;
      }
      if (adaptive) filter_adaptive_lane<RHS, q, EK1>(P, i0, (unsigned)(i - i0));
      else if (P.everystep && P.stagger == 7) filter_fixed_lane<RHS, q, EK1, true, true>(P, i0, (unsigned)(i - i0));  // lagged record stores
      else if (P.everystep) filter_fixed_lane<RHS, q, EK1, true>(P, i0, (unsigned)(i - i0));
      else filter_fixed_lane<RHS, q, EK1, false>(P, i0, (unsigned)(i - i0));
    }
  }
};
struct RunSmooth {
  const SmoothParams& P;
  int bcast_rows = 0;  // the DPP-broadcast row-team smoother (rows_smooth.h) instead of the default for this size
  template <int d, int q>
  void operator()() {
    if constexpr (d * (q + 1) <= 16) {
      if (bcast_rows) {
        std::vector<double> ws(RowsSmoother<d, q, false>::kLdsDoubles);
        for (long i = 0; i < P.N; ++i) {
          const RowsTeam tm{i, i, true, 0, ws.data(), nullptr};
          if (P.adaptive) {
            RowsSmoother<d, q, true> sm;
            sm.run(P, tm, (long)P.nsaved[i]);
          } else {
            RowsSmoother<d, q, false> sm;
            sm.run(P, tm, P.n_save);
          }
        }
        return;
      }
    }
    if constexpr (d * (q + 1) <= 12) {  // lane-per-trajectory smoother, lane-private memory = a plain array here
      constexpr int D = d * (q + 1);
      std::vector<double> x(D * (D + 1) / 2);
      for (long i = 0; i < P.N; ++i) {
        if (P.adaptive)
          smooth_lane_v2<d, q, true>(P, i, 0, LaneMem{x.data(), 1}, (long)P.nsaved[i]);
        else
          smooth_lane_v2<d, q, false>(P, i, 0, LaneMem{x.data(), 1}, P.n_save);
      }
    } else if constexpr (d * (q + 1) <= 32) {  // row-per-lane teams: all lanes of a team emulated phase by phase
      constexpr int D = d * (q + 1), TEAM = (D <= 16) ? 16 : 32;
      std::vector<double> ws(RowsWs<d, q + 1>::size);
      std::vector<RowState<D>> st(TEAM);
      for (long i = 0; i < P.N; ++i) smooth_rows_lane<d, q, TEAM>(P, i, 0, ws.data(), st.data());
    } else {
      std::vector<double> ws(SmoothWs<d, q + 1>::size);
      for (long i = 0; i < P.N; ++i) smooth_team_lane<d, q, 1>(P, i, 0, ws.data());
    }
  }
};

extern "C" int emul_filter(const EmulArgs* a) {
  FilterParams P;
  fill(*a, P);
  RunFilter r{P, a->adaptive};
  switch (a->rhs) {
    case 0: return dispatch_order<RhsFHN>(a->q, a->ek1, r);
    case 1: return dispatch_order<RhsLorenz63>(a->q, a->ek1, r);
    case 2: return dispatch_order<RhsLotkaVolterra>(a->q, a->ek1, r);
    case 3: return dispatch_order<RhsVanDerPol>(a->q, a->ek1, r);
    case 4: return dispatch_order<RhsLinear>(a->q, a->ek1, r);
    default: return -2;
  }
}

extern "C" int emul_smooth(const EmulArgs* a, int d) {
  SmoothParams P;
  std::memcpy(P.pc.At, a->At, sizeof(P.pc.At));
  std::memcpy(P.pc.Qt, a->Qt, sizeof(P.pc.Qt));
  std::memcpy(P.pc.QLt, a->QLt, sizeof(P.pc.QLt));
  P.N = a->N; P.n_save = a->n_save; P.adaptive = a->adaptive;
  P.hs = a->hs; P.ptab = a->ptab; P.tab_idx = a->tab_idx; P.tsave = a->tsave; P.nsaved = a->nsaved;
  P.mean = a->mean; P.cov = a->cov; P.diff = a->diff; P.smean = a->smean; P.scov = a->scov;
  P.retcode = a->retcode;
  RunSmooth r{P, a->everystep == 3};
  if (d == 2) return dispatch_smooth_order<2>(a->q, r);
  if (d == 3) return dispatch_smooth_order<3>(a->q, r);
  if (d == 28) return dispatch_smooth_order<28>(a->q, r);
  return -2;
}

// precond_fill of ek_math.h for the Python driver (same source as the product's host tables)
extern "C" void emul_precond_fill(int q, double h, double pval, double* tab) {
  switch (q) {
    case 1: precond_fill<2>(h, pval, tab); break;
    case 2: precond_fill<3>(h, pval, tab); break;
    case 3: precond_fill<4>(h, pval, tab); break;
    case 4: precond_fill<5>(h, pval, tab); break;
    case 5: precond_fill<6>(h, pval, tab); break;
  }
}
extern "C" int emul_tab_stride() { return kTabStride; }

// team (workgroup-per-trajectory) filter with TEAM = 1: Pleiades, and Lorenz-63 as a cross-check
struct RunTeamFilter {
  const FilterParams& P;
  template <class RHS, int q, bool EK1>
  void operator()() {
    using TF = TeamFilter<RHS, q, EK1, 1>;
    std::vector<double> ws((size_t)P.N * TF::W::size), sm(TF::W::small_size);
    TeamFilterParams TP{P, ws.data()};
    for (long i = 0; i < P.N; ++i) TF::run(TP, i, 0, sm.data());
  }
};
extern "C" int emul_filter_team(const EmulArgs* a) {
  FilterParams P;
  fill(*a, P);
  RunTeamFilter r{P};
  switch (a->rhs) {
    case 1: return dispatch_order<RhsLorenz63>(a->q, a->ek1, r);
    case 5: return dispatch_order<RhsPleiades>(a->q, a->ek1, r);
    default: return -2;
  }
}

// dense output (dense_lane.h)
struct EmulDense {
  const EmulArgs* a;
  int smoothed;
  const double* tq;
  long n_q;
  double *qmean, *qcov;
};
struct RunDense {
  const DenseParams& P;
  template <int d, int q>
  void operator()() {
    if constexpr (d * (q + 1) <= 12) {
      constexpr int D = d * (q + 1);
      std::vector<double> x(D * (D + 1) / 2);
      for (long j = 0; j < P.n_q; ++j)
        for (long i = 0; i < P.N; ++i) dense_lane<d, q>(P, i, j, LaneMem{x.data(), 1});
    }
  }
};
extern "C" int emul_dense(const EmulDense* e, int d) {
  const EmulArgs* a = e->a;
  DenseParams P;
  std::memset(&P, 0, sizeof P);
  std::memcpy(P.pc.At, a->At, sizeof(P.pc.At));
  std::memcpy(P.pc.Qt, a->Qt, sizeof(P.pc.Qt));
  std::memcpy(P.pc.QLt, a->QLt, sizeof(P.pc.QLt));
  P.N = a->N; P.n_save = a->n_save; P.adaptive = a->adaptive; P.smoothed = e->smoothed;
  P.tgrid = a->hs;  /* the Python driver passes the time grid here for dense output */
  P.tsave = a->tsave; P.nsaved = a->nsaved; P.mean = a->mean; P.cov = a->cov; P.diff = a->diff;
  P.smean = a->smean; P.scov = a->scov; P.tq = e->tq; P.n_q = e->n_q; P.qmean = e->qmean; P.qcov = e->qcov;
  RunDense r{P};
  if (d == 2) return dispatch_smooth_order<2>(a->q, r);
  if (d == 3) return dispatch_smooth_order<3>(a->q, r);
  return -2;
}

// posterior sampling (sample_lane.h)
struct EmulSample {
  const EmulArgs* a;
  long n_samples;
  unsigned long long seed;
  double noise_scale;
  double* samples;
};
struct RunSample {
  const SampleParams& P;
  template <int d, int q>
  void operator()() {
    if constexpr (d * (q + 1) <= 12) {
      constexpr int D = d * (q + 1);
      std::vector<double> x(D * (D + 1) / 2);
      for (long j = 0; j < P.n_samples; ++j)
        for (long i = 0; i < P.N; ++i) sample_lane<d, q>(P, i, j, LaneMem{x.data(), 1}, (P.adaptive && !P.tq) ? (long)P.nsaved[i] : P.n_save);
    }
  }
};
extern "C" int emul_sample(const EmulSample* e, int d) {
  const EmulArgs* a = e->a;
  SampleParams P;
  std::memset(&P, 0, sizeof P);
  std::memcpy(P.pc.At, a->At, sizeof(P.pc.At));
  std::memcpy(P.pc.Qt, a->Qt, sizeof(P.pc.Qt));
  std::memcpy(P.pc.QLt, a->QLt, sizeof(P.pc.QLt));
  P.N = a->N; P.n_save = a->n_save; P.adaptive = a->adaptive;
  P.hs = a->hs; P.ptab = a->ptab; P.tab_idx = a->tab_idx;
  P.tsave = a->tsave; P.nsaved = a->nsaved; P.mean = a->mean; P.cov = a->cov; P.diff = a->diff;
  P.n_samples = e->n_samples; P.seed = e->seed; P.noise_scale = e->noise_scale; P.samples = e->samples;
  RunSample r{P};
  if (d == 2) return dispatch_smooth_order<2>(a->q, r);
  if (d == 3) return dispatch_smooth_order<3>(a->q, r);
  return -2;
}

// dense-grid sampling (sample_lane.h, tq != nullptr): qmean/qcov hold the filter posterior at tq (emul_dense, smoothed = 0)
struct EmulDenseSample {
  const EmulArgs* a;
  const double* tq;
  long n_q;
  const double* qmean;
  const double* qcov;
  const double* rec_t;  // fixed solves: record times [n_save]
  long n_samples;
  unsigned long long seed;
  double noise_scale;
  double* samples;  // [n_q][D][n_samples][N]
};
extern "C" int emul_dense_sample(const EmulDenseSample* e, int d) {
  const EmulArgs* a = e->a;
  SampleParams P;
  std::memset(&P, 0, sizeof P);
  std::memcpy(P.pc.At, a->At, sizeof(P.pc.At));
  std::memcpy(P.pc.Qt, a->Qt, sizeof(P.pc.Qt));
  std::memcpy(P.pc.QLt, a->QLt, sizeof(P.pc.QLt));
  P.N = a->N; P.n_save = e->n_q; P.adaptive = a->adaptive;
  P.tsave = a->tsave; P.nsaved = a->nsaved; P.mean = e->qmean; P.cov = e->qcov; P.diff = a->diff;
  P.tq = e->tq; P.rec_t = e->rec_t; P.n_rec = a->n_save;
  P.n_samples = e->n_samples; P.seed = e->seed; P.noise_scale = e->noise_scale; P.samples = e->samples;
  RunSample r{P};
  if (d == 2) return dispatch_smooth_order<2>(a->q, r);
  if (d == 3) return dispatch_smooth_order<3>(a->q, r);
  return -2;
}

// register-tiled Pleiades filter (filter_tiles.h): all 320 tile threads emulated phase by phase, helper sections in line
struct RunTilesFilter {
  const FilterParams& P;
  int adaptive;
  template <class RHS, int q, bool EK1>
  void operator()() {
    using TF = TilesFilter<RHS, q, EK1>;
    std::vector<double> sm(TF::W::size);
    std::vector<TileState> st(kTilesThreads);
    for (long i = 0; i < P.N; ++i) {
      if (adaptive) TF::template run_adaptive<false>(P, i, 0, sm.data(), st.data());
      else TF::template run<false>(P, i, 0, sm.data(), st.data());
    }
  }
};
extern "C" int emul_filter_tiles(const EmulArgs* a) {
  FilterParams P;
  fill(*a, P);
  RunTilesFilter r{P, a->adaptive};
  if (a->rhs != 5) return -2;
  return dispatch_order<RhsPleiades>(a->q, a->ek1, r);
}
