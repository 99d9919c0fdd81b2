// RTS smoother for the workgroup-per-trajectory path (Pleiades, D = 28 (q+1) up to 168) ON THE MATRIX CORES:
// the whole step of src/smoothing.jl:31-63 is dense D x D algebra, and every O(D^3) part of it runs as
// v_mfma_f64_16x16x4_f64 products (csrc/mfma_dense.h):
//
//   X = P Sigma_i P                     unpack the packed record (full symmetric, padded to DP = 16 ceil(D/16))
//   Yt = A X,  B = Yt-rows A' + sigma^2 Q   the prior is A = At (x) I_d: block-row / block-column combinations, O(D^2 q)
//   M = P Sigma^s_{i+1} P - B
//   B = U'U                             blocked Cholesky, block row = MFMA product against the explicitly inverted 16 x 16
//                                       diagonal block (no per-row substitution)                    [D^3 / 3 flops]
//   Gt = (U'U)^-1 Yt                    two block sweeps, again only products                       [2 D^3]
//   m^s = m + G (m^s_{i+1} - A m)
//   Sigma^s = X + G M G'                two products Z = M Gt, R = Z' Gt (M symmetric)               [4 D^3]
//
// The first version of this smoother (csrc/smooth_team.h, out of the library; what the host
// emulation, tests/emul, runs) does the same algebra with 7 x 7 register tiles of vector FMAs and per-row substitutions out of
// a global workspace: 1.57e5 steps/s.  The textbook form X + G (S^s_+ - S^-) G' is the identity the reference's own test
// asserts for its stacked-QR Joseph form (test/filtering.jl:113).
//
// One workgroup of 256 threads (4 wavefronts) per trajectory; matrices in a per-trajectory global workspace
// (7 DP^2 doubles, L2 / Infinity-Cache resident while in use), vectors and the inverted diagonal blocks in LDS.
#pragma once
#include "ek_lane.h"
#include "mfma_dense.h"

namespace odef {

#ifdef ODEF_MFMA_STAMPS  // diagnostic build (tools/mfma_smooth_stamps.hip): cycles per phase of workgroup 0
__device__ unsigned long long g_mfma_stamps[16];
__device__ unsigned long long g_mfma_t0;
#define ODEF_STAMP(k)                                                          \
  do {                                                                         \
    __syncthreads();                                                           \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                 \
      const unsigned long long now_ = wall_clock64();                          \
      g_mfma_stamps[k] += now_ - g_mfma_t0;                                    \
      g_mfma_t0 = now_;                                                        \
    }                                                                          \
  } while (0)
#else
#define ODEF_STAMP(k)
#endif

template <int d, int NB>
struct MfmaSmoothWs {
  static constexpr int D = d * NB, DPB = (D + 15) / 16, DP = DPB * 16, MAT = DP * DP;
  static constexpr int X = 0, YT = MAT, BM = 2 * MAT, LM = 3 * MAT, MM = 4 * MAT, Z2 = 5 * MAT, SG = 6 * MAT;
  static constexpr int MSV = 7 * MAT;  // the carried smoothed mean between the launches of a staged pass
  // (the split pass -- smooth_predict.h, rts_smooth_sweeps_kernel -- uses YT, BM and SG of the matrices only, tile-major, see below)
  // split pass (one kernel per phase): what the phases of a record hand over -- P m, delta, P^-1 (vectors) and the record
  // index the hand-over belongs to (-1: that record needs no algebra)
  static constexpr int MFV = MSV + DP, DLV = MFV + DP, PIJV = DLV + DP, FLG = PIJV + DP, PJV = FLG + 8;  // (PJV: P)
  static constexpr size_t size = (size_t)PJV + DP;
  // ... where B, Y' and the carried Sigma^s lie TILE-MAJOR in their regions (BM, YT, SG; B and Sigma^s: the tiles on and above
  // the diagonal): 16 x 16 tiles of 256 contiguous doubles, the tiles of a
  // tile column one after the other -- what one wavefront of rts_smooth_sweeps_kernel loads is contiguous (a tile 2 KB, a tile
  // column of Y' 2 KB x DPB) instead of 128-byte pieces DP doubles apart
  __host__ __device__ static constexpr int tile_at(int tr, int tc) { return (tc * DPB + tr) * 256; }
  __host__ __device__ static constexpr int tm(int r, int c) { return tile_at(r >> 4, c >> 4) + (r & 15) * 16 + (c & 15); }
  // LDS (doubles): factorisation scratch, then the vectors
  static constexpr int kChol = mf::CholLds<DPB>::size;
  static constexpr int MF = kChol, MS = MF + DP, MP = MS + DP, DL = MP + DP, PJ = DL + DP, PIJ = PJ + DP;
  static constexpr int lds_size = PIJ + DP;
};

// (a, b) of packed lower-triangle element e, advanced by `step` elements at a time without square roots
struct TriWalk {
  int a, b;
  __device__ inline TriWalk(int e) {
    a = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while (a * (a + 1) / 2 > e) --a;
    while ((a + 1) * (a + 2) / 2 <= e) ++a;
    b = e - a * (a + 1) / 2;
  }
  __device__ inline void advance(int step) {
    b += step;
    while (b > a) {
      b -= a + 1;
      ++a;
    }
  }
};

// next tile (tr, tc), tc <= tr, of the lower triangle in row order
__device__ __attribute__((always_inline)) inline void tri_next(int& tr, int& tc) {
  if (++tc > tr) {
    tc = 0;
    ++tr;
  }
}

// ---- the two halves of one RTS step on the workspace (shared by the smoother loop and the dense output, dense_mfma.h).
// In:  X = P Sigma P (full symmetric, preconditioned filter covariance), mf_ = P m, SG = Sigma^s_+ (un-preconditioned, full),
//      ms_ = m^s_+ (un-preconditioned), pj_ / pij_ = diag of P / P^-1 per state component.
// mfma_predict_phase:  Yt = A X, mp_ = A mf_, BM = A X A' + sigma^2 Q (the predicted covariance, preconditioned),
//                      MM = P Sigma^s_+ P - BM, dl_ = P m^s_+ - mp_
// mfma_gain_phase:     BM = U'U, Yt <- G' = BM^-1 Yt, ms_ <- P^-1 (mf_ + G dl_), BM <- G M G'   (returns "NaN seen")
template <int d, int q>
__device__ __attribute__((always_inline)) inline void mfma_predict_phase(const PriorConsts& pc, double sigma2, double* __restrict__ ws, double* __restrict__ lds) {
  using W = MfmaSmoothWs<d, q + 1>;
  constexpr int NB = q + 1, D = W::D, DP = W::DP, DPB = W::DPB;
  const int tid = (int)threadIdx.x, nth = (int)blockDim.x;
  double* X = ws + W::X;
  double* YT = ws + W::YT;
  double* BM = ws + W::BM;
  double* LM = ws + W::LM;
  double* MM = ws + W::MM;
  double* Z2 = ws + W::Z2;
  double* SG = ws + W::SG;
  double* mf_ = lds + W::MF;
  double* ms_ = lds + W::MS;
  double* mp_ = lds + W::MP;
  double* dl_ = lds + W::DL;
  double* pj_ = lds + W::PJ;
  double* pij_ = lds + W::PIJ;
  (void)X; (void)YT; (void)BM; (void)LM; (void)MM; (void)Z2; (void)SG; (void)mf_; (void)ms_; (void)mp_; (void)dl_; (void)pj_; (void)pij_; (void)DPB; (void)NB;
  // Yt = A X (row (J, a) picks up the rows (j, a), j > J);  m^- = A m~ (src/filtering.jl:22-25)
  // one work item = (component a, column c): the NB rows (j, a) of X in that column give all NB rows (J, a) of Yt
  for (int e = tid; e < d * DP; e += nth) {
    const int a = e / DP, c = e % DP;
    double x[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) x[j] = X[(j * d + a) * DP + c];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      double t = x[J];
#pragma unroll
      for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * x[j];
      YT[(J * d + a) * DP + c] = t;
    }
  }
  for (int k = tid; k < D; k += nth) {
    const int J = k / d, a = k % d;
    double t = mf_[k];
    for (int j = J + 1; j < NB; ++j) t += pc.At[J][j] * mf_[j * d + a];
    mp_[k] = t;
  }
  __syncthreads();
  ODEF_STAMP(1);  // Yt
  // B = A X A' + sigma^2 Q (src/filtering.jl:34-35) from the rows of Yt;  M = P Sigma^s_+ P - B;  delta
  // one work item = (row r, component b): the NB entries (k, b) of row r of Yt give all NB entries (K, b) of row r of B
  for (int e = tid; e < D * d; e += nth) {
    const int r = e / d, b = e % d;
    double yv[NB], sg[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      yv[k] = YT[r * DP + k * d + b];
      sg[k] = SG[r * DP + k * d + b];
    }
    const bool diag = (r % d) == b;
    const int J = r / d;
#pragma unroll
    for (int K = 0; K < NB; ++K) {
      double bv = yv[K];
#pragma unroll
      for (int k = K + 1; k < NB; ++k) bv += pc.At[K][k] * yv[k];
      double qv = 0.0;
#pragma unroll
      for (int JJ = 0; JJ < NB; ++JJ) qv = (JJ == J) ? pc.Qt[JJ][K] : qv;
      if (diag) bv += sigma2 * qv;
      const int c = K * d + b;
      BM[r * DP + c] = bv;
      MM[r * DP + c] = sg[K] * (pj_[r] * pj_[c]) - bv;
    }
  }
  // the padding block of B is the identity (R of the previous step left zeros there); M is zero there from the start
  for (int e = tid; e < (DP - D) * DP; e += nth) {
    const int r = D + e / DP, c = e % DP;
    BM[r * DP + c] = (r == c) ? 1.0 : 0.0;
    BM[c * DP + r] = (r == c) ? 1.0 : 0.0;
  }
  for (int k = tid; k < D; k += nth) dl_[k] = pj_[k] * ms_[k] - mp_[k];
  __syncthreads();
  ODEF_STAMP(2);  // B, M
}
// PART: 7 = all of it; 1 = the Cholesky only; 4 = what follows the sweeps (mean, products) -- the split pass runs the sweeps
// (2) in a kernel of their own
template <int d, int q, int PART = 7>
__device__ __attribute__((always_inline)) inline bool mfma_gain_phase(double* __restrict__ ws, double* __restrict__ lds) {
  using W = MfmaSmoothWs<d, q + 1>;
  constexpr int NB = q + 1, D = W::D, DP = W::DP, DPB = W::DPB;
  const int tid = (int)threadIdx.x, nth = (int)blockDim.x;
  double* X = ws + W::X;
  double* YT = ws + W::YT;
  double* BM = ws + W::BM;
  double* LM = ws + W::LM;
  double* MM = ws + W::MM;
  double* Z2 = ws + W::Z2;
  double* SG = ws + W::SG;
  double* mf_ = lds + W::MF;
  double* ms_ = lds + W::MS;
  double* mp_ = lds + W::MP;
  double* dl_ = lds + W::DL;
  double* pj_ = lds + W::PJ;
  double* pij_ = lds + W::PIJ;
  (void)X; (void)YT; (void)BM; (void)LM; (void)MM; (void)Z2; (void)SG; (void)mf_; (void)ms_; (void)mp_; (void)dl_; (void)pj_; (void)pij_; (void)DPB; (void)NB;
  bool nan_seen = false;
  // B = U'U, Gt = B^-1 Yt (the gain G = X A' (Sigma^-)^-1, src/smoothing.jl:42-43, transposed)
  if (PART & 1) {
    mf::wg_cholesky_upper<DPB>(BM, LM, DP, lds);
    ODEF_STAMP(3);  // Cholesky
  }
  if (PART == 1) return false;
  if (PART & 2) {
#ifndef ODEF_SMOOTH_RR
  mf::wg_solve_upper<DPB>(BM, LM, YT, DP, lds);  // left-looking: 22 block steps, a barrier after each
#else  // A/B build: right-hand sides resident in the accumulators, no barrier inside -- measured 1.5x SLOWER (see mfma_dense.h)
  mf::wg_solve_upper_rr<DPB>(BM, LM, YT, DP, lds);
  __syncthreads();
#endif
  ODEF_STAMP(4);  // sweeps
  }
  // m^s = m + G delta (src/smoothing.jl:44), un-preconditioned (:26)
  // (four running sums over the rows r = 0, 1, 2, 3 mod 4, combined as (s0 + s1) + (s2 + s3): the order in which the on-chip
  // kernel of the split pass, oc::gt_times, sums the same terms from its accumulator tiles -- un-preconditioning amplifies
  // rounding differences by h^-j in derivative block j, and the two passes are compared with each other)
  for (int k = tid; k < D; k += nth) {
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
    for (int r = 0; r < D; r += 4) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (D % 4 == 0 || r + g < D) s4[g] = fma(YT[(r + g) * DP + k], dl_[r + g], s4[g]);
    }
    const double v = (mf_[k] + ((s4[0] + s4[1]) + (s4[2] + s4[3]))) * pij_[k];
    nan_seen = nan_seen || !(v == v);
    ms_[k] = v;
  }
  // Z = M Gt, R = Z' Gt = G M G'
  ODEF_STAMP(5);  // mean
#ifndef ODEF_SMOOTH_RR
  mf::wg_atb<false>(MM, DP, YT, DP, DP, nullptr, Z2, DP, 0, DPB, 0, DPB);
#else
  mf::wg_atb_rescols<DPB>(MM, YT, Z2, DP);
#endif
  __syncthreads();
  ODEF_STAMP(6);  // Z = M Gt
#ifndef ODEF_SMOOTH_RR
  mf::wg_atb<false>(Z2, DP, YT, DP, DP, nullptr, BM, DP, 0, DPB, 0, DPB);
#else
  mf::wg_atb_rescols<DPB>(Z2, YT, BM, DP);
#endif
  __syncthreads();
  ODEF_STAMP(7);  // R = Z' Gt
  return nan_seen;
}

// SPLITK: the instantiation for the split pass (P.split_mode 1 / 2 only) -- a kernel of its own, so that neither carries the
// other's code and registers
template <int d, int q, bool SPLITK = false>
__device__ __attribute__((always_inline)) inline void smooth_mfma_traj(const SmoothParams& P, long i, double* __restrict__ ws, double* __restrict__ lds) {
  constexpr int NB = q + 1;
  using W = MfmaSmoothWs<d, NB>;
  constexpr int D = W::D, DP = W::DP, DPB = W::DPB, TRI = D * (D + 1) / 2;
  const int tid = (int)threadIdx.x, nth = (int)blockDim.x;
  const long n = P.adaptive ? (long)P.nsaved[i] : P.n_save;
  const size_t N = (size_t)P.N;
  const PriorConsts& pc = P.pc;
  double* X = ws + W::X;
  double* YT = ws + W::YT;
  double* BM = ws + W::BM;
  double* LM = ws + W::LM;
  double* MM = ws + W::MM;
  double* Z2 = ws + W::Z2;
  double* SG = ws + W::SG;
  double* mf_ = lds + W::MF;
  double* ms_ = lds + W::MS;
  double* mp_ = lds + W::MP;
  double* dl_ = lds + W::DL;
  double* pj_ = lds + W::PJ;
  double* pij_ = lds + W::PIJ;

  // Staged pass (fixed grids, api.hip): the covariance records of this launch lie trajectory-major in P.stage -- one record is
  // stage_ld contiguous doubles, read and written as whole lines -- instead of as 8-byte pieces N doubles apart.
  const bool staged = P.stage != nullptr;
  const long s_hi = staged ? (n - 2 < P.stage_hi ? n - 2 : P.stage_hi) : n - 2, s_lo = staged ? (P.stage_s0 > 1 ? P.stage_s0 : 1) : 1;
  auto rec = [&](long s) -> double* { return P.stage + ((size_t)(s - P.stage_s0) * N + (size_t)i) * (size_t)P.stage_ld; };
  if (staged && P.split_mode != 2) {
    // save slots this trajectory never used (adaptive solves: n differs between trajectories) leave the stage as zeros,
    // as the in-place pass leaves them
    for (long s = (n > P.stage_s0 ? n : P.stage_s0); s <= P.stage_hi; ++s) {
      double* dst = rec(s);
      for (int e = tid; e < (int)P.stage_ld; e += nth) dst[e] = 0.0;
    }
  }
  if (staged && n - 1 < P.stage_s0) return;  // (workgroup-uniform) its records all lie below this block of the stage
  const bool resume = staged && (n - 1 > P.stage_hi || P.split_mode == 2);  // started in an earlier launch
  if (!resume) {
    // zero the workspace once (padding rows / columns stay zero from here on); L = U' must be zero above the diagonal
    for (size_t e = tid; e < W::size; e += nth) ws[e] = 0.0;
    __syncthreads();
    // first and last record are copied (index 1 in Julia is never smoothed, src/smoothing.jl:11); the last one is the
    // carried smoothed state Sigma^s (SG, full symmetric, un-preconditioned)
    for (int w = 0; w < 2; ++w) {
      const long s = w == 0 ? 0 : n - 1;
      for (int k = tid; k < D; k += nth) {
        const double v = P.mean[((size_t)s * D + k) * N + i];
        P.smean[((size_t)s * D + k) * N + i] = v;
        ms_[k] = v;
      }
      if (staged) {  // the two covariance records are copied by the host; the last one is staged for this read
        if (w == 0) continue;
        const double* src = rec(s);
        TriWalk tw(tid);
        for (int e = tid; e < TRI; e += nth, tw.advance(nth)) {
          const double v = src[e];
          if constexpr (SPLITK) {  // the split pass carries Sigma^s as upper tiles, tile-major (W::tm), diagonal tiles complete
            SG[W::tm(tw.b, tw.a)] = v;
            if ((tw.a >> 4) == (tw.b >> 4)) SG[W::tm(tw.a, tw.b)] = v;
          } else {
            SG[tw.a * DP + tw.b] = v;
            SG[tw.b * DP + tw.a] = v;
          }
        }
        continue;
      }
      TriWalk tw(tid);
      for (int e = tid; e < TRI; e += nth, tw.advance(nth)) {
        const double v = P.cov[((size_t)s * TRI + e) * N + i];
        P.scov[((size_t)s * TRI + e) * N + i] = v;
        SG[tw.a * DP + tw.b] = v;
        SG[tw.b * DP + tw.a] = v;
      }
    }
  } else {
    for (int k = tid; k < D; k += nth) ms_[k] = ws[W::MSV + k];
  }
  __syncthreads();
  bool nan_seen = false;
  // One record in two parts with the sweeps between them: the persistent loop runs A, sweeps, C back to back; the split
  // pass (P.split_mode == 2) runs C of the previous record and A of the next one here and the sweeps in a kernel of their own.
  // part A: unpack, predict (and, in the persistent loop, what follows).  Returns false for a repeated save time (src/smoothing.jl:13-16: the smoothed state
  // carries over, nothing to factorise).
  auto part_a = [&](long s, auto split) -> bool {
    constexpr bool SPLIT = decltype(split)::value;
    double h;
    if (P.adaptive) h = P.tsave[(size_t)(s + 1) * N + i] - P.tsave[(size_t)s * N + i];
    else h = uniform_load(P.hs + s);
    if (h == 0.0) {  // src/smoothing.jl:13-16: a repeated save time, the smoothed state carries over
      for (int k = tid; k < D; k += nth) P.smean[((size_t)s * D + k) * N + i] = ms_[k];
      if (staged) {
        double* dst = rec(s);
        TriWalk tw(tid);
        for (int e = tid; e < TRI; e += nth, tw.advance(nth)) dst[e] = SG[tw.a * DP + tw.b];
      } else {
        for (int e = tid; e < TRI; e += nth) P.scov[((size_t)s * TRI + e) * N + i] = P.scov[((size_t)(s + 1) * TRI + e) * N + i];
      }
      __syncthreads();
      return false;
    }
    // preconditioner of this step (src/preconditioning.jl:1-17), per state component
    if (tid < DP) {
      double pj = 0.0, pij = 0.0;
      if (tid < D) {
        if (P.adaptive) {
          double tabv[kTabStride];
          precond_table_fast<q, NB>(h, tabv);
          pj = tabv[kTabPJ + tid / d];
          pij = tabv[kTabPIJ + tid / d];
        } else {
          const GlobalTab tab{P.ptab + (size_t)uniform_load(P.tab_idx + s) * kTabStride};
          pj = tab[kTabPJ + tid / d];
          pij = tab[kTabPIJ + tid / d];
        }
      }
      pj_[tid] = pj;
      pij_[tid] = pij;
    }
    const double sigma2 = P.diff[(size_t)(s + 1) * N + i];
    __syncthreads();
    // X = P Sigma_i P (src/smoothing.jl:23), m~ = P m_i
    if (staged) {
      // by 16 x 16 tiles of the lower triangle: a tile row is 128 contiguous bytes of the record, and the mirrored tile
      // goes out as 32-byte pieces that complete their lines within four stores (element by element the mirror image
      // was 14 196 scattered 8-byte writes per step)
      const double* src = rec(s);
      const int wave = tid >> 6, nw = nth >> 6, li = (tid & 63) >> 4, lj = tid & 15;
      int tr = 0, tc = 0;
      for (int t = 0; t < wave; ++t) tri_next(tr, tc);
      for (int t = wave; t < DPB * (DPB + 1) / 2; t += nw) {
        mf::d4 x;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int a = tr * 16 + 4 * v + li, b = tc * 16 + lj;
          const int hi = a > b ? a : b, lo = a > b ? b : a;  // (a diagonal tile is read as the full symmetric block)
          x[v] = hi < D ? src[hi * (hi + 1) / 2 + lo] * (pj_[a] * pj_[b]) : 0.0;
        }
        mf::store_tile(X, DP, tr * 16, tc * 16, x);
        if (tr != tc) mf::store_tile_t(X, DP, tc * 16, tr * 16, x);
        for (int k = 0; k < nw; ++k) tri_next(tr, tc);
      }
    } else {
      TriWalk tw(tid);
      const double* src = P.cov + ((size_t)s * TRI) * N + i;
#pragma unroll 4
      for (int e = tid; e < TRI; e += nth, tw.advance(nth)) {
        const double v = src[(size_t)e * N] * (pj_[tw.a] * pj_[tw.b]);
        X[tw.a * DP + tw.b] = v;
        X[tw.b * DP + tw.a] = v;
      }
    }
    for (int k = tid; k < D; k += nth) mf_[k] = pj_[k] * P.mean[((size_t)s * D + k) * N + i];
    __syncthreads();
    ODEF_STAMP(0);  // unpack
    mfma_predict_phase<d, q>(pc, sigma2, ws, lds);
    static_assert(!SPLIT, "the split pass unpacks and predicts in rts_smooth_predict_kernel (smooth_predict.h)");
    return true;
  };
  // part C: smoothed mean, the two products, pack and store
  auto part_c = [&](long s, auto split) {
    constexpr bool SPLIT = decltype(split)::value;
    static_assert(!SPLIT, "the split pass finishes its records in rts_smooth_sweeps_kernel");
    for (int k = tid; k < D; k += nth) P.smean[((size_t)s * D + k) * N + i] = ms_[k];
    // Sigma^s = P^-1 (X + G M G') P^-1: the record (packed lower triangle) and the carried full matrix
    if (staged) {  // by tiles, as the unpacking above; the lower triangle of the sum is what both halves of SG get
      double* dst = rec(s);
      const int wave = tid >> 6, nw = nth >> 6, li = (tid & 63) >> 4, lj = tid & 15;
      int tr = 0, tc = 0;
      for (int t = 0; t < wave; ++t) tri_next(tr, tc);
      for (int t = wave; t < DPB * (DPB + 1) / 2; t += nw) {
        const mf::d4 x = mf::load_tile(X, DP, tr * 16, tc * 16), r = mf::load_tile(BM, DP, tr * 16, tc * 16);
        mf::d4 o;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int a = tr * 16 + 4 * v + li, b = tc * 16 + lj;
          o[v] = (x[v] + r[v]) * (pij_[a] * pij_[b]);
          if (a < D && b <= a) dst[a * (a + 1) / 2 + b] = o[v];
        }
        if (tr != tc) {
          mf::store_tile(SG, DP, tr * 16, tc * 16, o);
          mf::store_tile_t(SG, DP, tc * 16, tr * 16, o);
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int a = tr * 16 + 4 * v + li, b = tc * 16 + lj;
            if (b <= a) {
              SG[a * DP + b] = o[v];
              SG[b * DP + a] = o[v];
            }
          }
        }
        for (int k = 0; k < nw; ++k) tri_next(tr, tc);
      }
    } else {
      TriWalk tw(tid);
      double* dst = P.scov + ((size_t)s * TRI) * N + i;
#pragma unroll 4
      for (int e = tid; e < TRI; e += nth, tw.advance(nth)) {
        const double v = (X[tw.a * DP + tw.b] + BM[tw.a * DP + tw.b]) * (pij_[tw.a] * pij_[tw.b]);
        dst[(size_t)e * N] = v;
        SG[tw.a * DP + tw.b] = v;
        SG[tw.b * DP + tw.a] = v;
      }
    }
    __syncthreads();
    ODEF_STAMP(8);  // pack + store
    };
  if constexpr (SPLITK) {
    // the split pass: this launch only sets up / carries over the block's state (above); its records run as
    // [rts_smooth_predict_kernel -> rts_smooth_sweeps_kernel] pairs, which hand over through the workspace (W::FLG: the
    // record index the first one prepared, -1: none)
    if (tid == 0) ws[W::FLG] = -1.0;
  } else {
    for (long s = s_hi; s >= s_lo; --s) {
      if (!part_a(s, std::false_type{})) continue;
      nan_seen = mfma_gain_phase<d, q>(ws, lds) || nan_seen;
      part_c(s, std::false_type{});
    }
  }
  if (staged) {
    for (int k = tid; k < D; k += nth) ws[W::MSV + k] = ms_[k];
  }
  if (nan_seen) P.retcode[i] = 3;  // "NaNs after smoothing" (src/smoothing.jl:25)
}

}  // namespace odef
