// Record stores of the 16-lanes-per-trajectory kernels.
//
// A record element of one save slot is a row of N doubles ([slot][element][N], include/odefilter.h).  A wavefront of the
// row-team kernels holds 4 trajectories, so stored directly every element would be a 32-byte piece of a 128-byte
// line -- and partial-line writes reach 0.7-0.8 TB/s on this chip where full lines reach 5.7 (tools/rows_store_bench.hip,
// profiles/r02_rows_store_shapes.txt): at 16 384 trajectories the every-step filter was STORE-bound at 17 ms.
// Therefore a workgroup is FOUR wavefronts = 16 consecutive trajectories = exactly one 128-byte line per element, and
// the record leaves through LDS:
//   1. the covariance is ALREADY in LDS: the step ends with the symmetrisation exchange, which leaves row r of the new
//      (un-preconditioned) covariance of every team in the team's exchange rows; mean, diffusion and time go into a small
//      staging image [element][16],
//   2. s_barrier,
//   3. each wavefront stores a quarter of the record: one instruction = 4 elements x 16 trajectories = 4 full lines,
//      read from wherever the element lives (exchange rows of team tt, lower triangle; staging image),
//   4. s_barrier: nobody overwrites its exchange rows (the next step's first exchange) before everybody has read them.
// put() is therefore a COLLECTIVE of the workgroup: every wavefront calls it the same number of times (teams with
// nothing to store -- a finished or not yet started trajectory, the padding of the last workgroup -- say so and are
// masked per trajectory on the store side).  It returns the OR of the teams' `more` flags, which is what the adaptive
// loops use as their workgroup-uniform continuation test.
// (First version: the whole record was copied into a double-buffered staging image; its 14 LDS writes per step, bank
// conflicted at a pitch of 16 doubles, cost three times what the global stores cost -- profiles/r02_rows_store_knobs.txt.)
//
// Host emulation (tests/emul): one team at a time, direct stores.
#pragma once
#include "team_vec.h"

namespace odef {

constexpr int kRowsWgTeams = 16;  // teams (trajectories) per workgroup: 4 wavefronts x 4
// Row pitch (doubles) of the staging image [element][16 trajectories].  NOT 16: the lanes of a team write elements
// tri(r, c), r = 0..15, i.e. rows that are whole multiples of the pitch apart, and with a pitch of 128 bytes all of them
// fall into the same few LDS banks -- measured: the 14 staging writes of a record cost 0.64 us per step, three times
// the global stores they feed (profiles/r02_rows_store_knobs.txt).  17 spreads the rows over the banks.
constexpr int kStagePitch = 17;

struct RowsTeam {     // what a team knows about its place
  long i;             // trajectory (clamped into range for the padding teams of the last workgroup)
  long i16;           // first trajectory of the workgroup
  bool valid;         // i is a real trajectory
  int tcol;           // team index inside the workgroup, 0..15
  double* lds_team;   // the team's exchange rows (team_vec.h)
  double* stage;      // the workgroup's staging image (device only)
};

// Fields of one record kind: MEAN [D rows], COV_TRIL [TRI rows], optionally one scalar row each for DIFFUSION and T.
template <int D, bool HAS_DIFF, bool HAS_T>
struct RowsSink {
  static constexpr int TRI = D * (D + 1) / 2;
  static constexpr int REC = D + TRI + (HAS_DIFF ? 1 : 0) + (HAS_T ? 1 : 0);  // elements of a record
  // staging image (doubles): mean rows, the scalar rows, the teams' flags, a dump row for masked lanes
  static constexpr int kStageRows = D + 2 + 2;
  static constexpr int kStageDoubles = kStageRows * kStagePitch;
  double* f_mean;
  double* f_cov;
  double* f_diff;
  double* f_t;
  size_t N;

#ifdef ODEF_HOST_EMUL
  long i;
  inline void init(const RowsTeam& tm, long N_, int, int, double* mean, double* cov, double* diff, double* t) {
    N = (size_t)N_;
    i = tm.i;
    f_mean = mean; f_cov = cov; f_diff = diff; f_t = t;
  }
  // xr: the team's covariance rows (on the device they are read from the team's exchange rows, where the step's
  // symmetrisation left them)
  inline bool put(long slot, bool store, bool more, const tv::TV& m, const tv::TV (&xr)[D], double diffusion, double t) {
    if (store) {
      for (int r = 0; r < D; ++r) {
        f_mean[((size_t)slot * D + r) * N + i] = m.v[r];
        for (int c = 0; c <= r; ++c) f_cov[((size_t)slot * TRI + tri(r, c)) * N + i] = xr[c].v[r];
      }
      if constexpr (HAS_DIFF) f_diff[(size_t)slot * N + i] = diffusion;
      if constexpr (HAS_T) f_t[(size_t)slot * N + i] = t;
    }
    return more;
  }
#else
  static constexpr int NJ = (REC + 3) / 4;  // store instructions per record (4 elements each)
  static constexpr int NK = (NJ + 3) / 4;   // ... per wavefront
  double* stage;
  int so_mean, so_one;  // the lane's staging slots (masked lanes point at the dump row)
  int tcol;
  // store side: instruction k of this wavefront covers elements 4 j .. 4 j + 3, j = 4 k + wave; this lane holds
  // (element e = 4 j + lane / 16, trajectory tt = lane % 16)
  double* g_base[NK];       // address of [slot 0][element e][i16 + tt] in its field (nullptr: nothing to store)
  unsigned g_stride[NK];    // bytes from one slot of that field to the next
  const double* rd_src[NK]; // where (e, tt) lives in LDS

  // lds_teams: the workgroup's exchange rows (team tt at lds_teams + tt * team_doubles, row pitch LD)
  __device__ inline void init(const RowsTeam& tm, long N_, int team_doubles, int LD, double* mean, double* cov, double* diff, double* t) {
    N = (size_t)N_;
    f_mean = mean; f_cov = cov; f_diff = diff; f_t = t;
    stage = tm.stage;
    tcol = tm.tcol;
    const double* lds_teams = tm.lds_team - tm.tcol * team_doubles;
    const int r = tv::lane();
    const int dump = (D + 3) * kStagePitch + tcol;
    so_mean = r < D ? r * kStagePitch + tcol : dump;
    so_one = r == 0 ? D * kStagePitch + tcol : dump;
    const int wave = (int)(threadIdx.x / 64u), lane64 = (int)(threadIdx.x % 64u);
    const int tt = lane64 % kRowsWgTeams;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int e = 4 * (4 * k + wave) + lane64 / kRowsWgTeams;
      const bool in = e < REC && tm.i16 + tt < N_;
      double* fb = nullptr;
      long row = 0, rows = 1;
      const double* src = stage;
      if (e < D) {
        fb = f_mean; row = e; rows = D;
        src = stage + e * kStagePitch + tt;
      } else if (e < D + TRI) {
        fb = f_cov; row = e - D; rows = TRI;
        int rr = 0;  // (rr, cc) with tri(rr, cc) == e - D
        while ((rr + 1) * (rr + 2) / 2 <= e - D) ++rr;
        const int cc = e - D - rr * (rr + 1) / 2;
        src = lds_teams + tt * team_doubles + rr * LD + cc;
      } else if (HAS_DIFF && e == D + TRI) {
        fb = f_diff;
        src = stage + D * kStagePitch + tt;
      } else {
        fb = f_t;
        src = stage + (D + 1) * kStagePitch + tt;
      }
      g_base[k] = in ? fb + ((size_t)row * N + (size_t)(tm.i16 + tt)) : nullptr;
      g_stride[k] = (unsigned)((size_t)rows * N * sizeof(double));
      rd_src[k] = src;
    }
  }
  // COLLECTIVE.  store: this team has a record for `slot`;  more: this team wants another round.
  // The team's covariance rows must be in its exchange rows (RowsStep::run / RowsSmoother leave them there); a team
  // that stores a state it did not just compute calls stage_cov() first.
  __device__ inline bool put(long slot, bool store, bool more, const tv::TV& m, const tv::TV (&)[D], double diffusion, double t) {
    if (store) {
      stage[so_mean] = m;
      if constexpr (HAS_DIFF) stage[so_one] = diffusion;
      if constexpr (HAS_T) stage[so_one + kStagePitch] = t;
    }
    int* flags = (int*)(stage + (D + 2) * kStagePitch);  // one 8-byte cell per team
    if (tv::lane() == 0) flags[2 * tcol] = (store ? 1 : 0) | (more ? 2 : 0);
    __syncthreads();
    const int fl = flags[2 * (int)(threadIdx.x % kRowsWgTeams)];  // the flags of trajectory tt
    const bool any_more = __builtin_amdgcn_ballot_w64((fl & 2) != 0) != 0ull;
    double v[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) v[k] = *rd_src[k];  // all reads in flight before the first store
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if (g_base[k] != nullptr && (fl & 1)) {
        double* dst = (double*)((char*)g_base[k] + (size_t)slot * g_stride[k]);
        __builtin_nontemporal_store(v[k], dst);
      }
    }
    __syncthreads();  // exchange rows and staging image are free again
    return any_more;
  }
#endif
  // put the rows of a state that is NOT the one the last exchange left in LDS (initial record, a restored state after
  // a rejected step, a carried smoothed state) into the team's exchange rows
  __device__ inline void stage_cov(const tv::Lds& lds, int LD, const tv::TV (&xr)[D]) const {
    tv::lds_put_row<D>(lds, LD, xr);
    tv::lds_sync();
  }
};

}  // namespace odef
