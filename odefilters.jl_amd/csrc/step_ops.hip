// Step-level entry points odef_predict / odef_update / odef_smooth_step: the pure
// functions predict!, update!, smooth of src/filtering.jl:17-48,79-91,136-154 on batches of
// Gaussians with run-time dimension D <= ODEF_MAX_STEP_DIM.  One lane per instance,
// per-instance workspace in global memory.  These exist for unit parity with
// test/filtering.jl; the time loop uses the register-resident kernels in ek_lane.h.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/odefilter.h"

namespace {

struct Mat {  // row-major view
  double* p;
  int ld;
  __device__ double& operator()(int i, int j) const { return p[i * ld + j]; }
};

// in-place Cholesky (lower) with the semidefinite rule of ek_math.h::chol_packed
__device__ void chol_lower(Mat B, int n) {
  for (int k = 0; k < n; ++k) {
    const double piv = B(k, k);
    const bool ok = piv > 0.0;
    const double lkk = ok ? sqrt(piv) : 0.0;
    const double inv = ok ? 1.0 / lkk : 0.0;
    B(k, k) = lkk;
    for (int i = k + 1; i < n; ++i) B(i, k) *= inv;
    for (int j = k + 1; j < n; ++j) {
      const double ljk = B(j, k);
      for (int i = j; i < n; ++i) B(i, j) -= B(i, k) * ljk;
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) B(i, j) = 0.0;
}

// X (n x n) <- solve (Lc Lc') X' = Y'  row-wise, i.e. X = Y (Lc Lc')^-1 for symmetric Lc Lc'
__device__ void solve_right_spd(Mat Lc, int n, Mat Y, int rows) {
  for (int r = 0; r < rows; ++r) {
    for (int k = 0; k < n; ++k) {
      double t = Y(r, k);
      for (int c = 0; c < k; ++c) t -= Lc(k, c) * Y(r, c);
      Y(r, k) = (Lc(k, k) != 0.0) ? t / Lc(k, k) : 0.0;
    }
    for (int k = n - 1; k >= 0; --k) {
      double t = Y(r, k);
      for (int c = k + 1; c < n; ++c) t -= Lc(c, k) * Y(r, c);
      Y(r, k) = (Lc(k, k) != 0.0) ? t / Lc(k, k) : 0.0;
    }
  }
}

// B = [A*L  Q_L][A*L  Q_L]'  then Cholesky (src/filtering.jl:33-41)
__device__ void predict_factor(int D, const double* A, const double* QL, const double* L, Mat M, Mat B) {
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) {
      double t = 0.0;
      for (int k = 0; k < D; ++k) t += A[i * D + k] * L[k * D + j];
      M(i, j) = t;
    }
  for (int i = 0; i < D; ++i)
    for (int j = 0; j <= i; ++j) {
      double t = 0.0;
      for (int k = 0; k < D; ++k) t += M(i, k) * M(j, k) + QL[i * D + k] * QL[j * D + k];
      B(i, j) = t;
      B(j, i) = t;
    }
  chol_lower(B, D);
}

__device__ void gram_out(int D, Mat F, int cols, double* out) {
  for (int i = 0; i < D; ++i)
    for (int j = 0; j <= i; ++j) {
      double t = 0.0;
      for (int k = 0; k < cols; ++k) t += F(i, k) * F(j, k);
      out[i * D + j] = t;
      out[j * D + i] = t;
    }
}

__global__ void predict_kernel(int D, long n, const double* mu, const double* L, const double* A, const double* QL,
                               double* mu_out, double* cov_out, double* ws) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double* w = ws + (size_t)i * 2 * D * D;
  Mat M{w, D}, B{w + D * D, D};
  predict_factor(D, A, QL, L + (size_t)i * D * D, M, B);
  for (int r = 0; r < D; ++r) {
    double t = 0.0;
    for (int k = 0; k < D; ++k) t += A[r * D + k] * mu[(size_t)i * D + k];
    mu_out[(size_t)i * D + r] = t;
  }
  gram_out(D, B, D, cov_out + (size_t)i * D * D);
}

__global__ void update_kernel(int D, int o, long n, const double* mu, const double* L, const double* H, const double* z,
                              double* mu_out, double* cov_out, double* ws) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* Li = L + (size_t)i * D * D;
  const double* Hi = H + (size_t)i * o * D;
  double* w = ws + (size_t)i * (2 * o * D + o * o + D * D);
  Mat HL{w, D}, K{w + o * D, o}, S{w + 2 * o * D, o}, Lf{w + 2 * o * D + o * o, D};
  for (int r = 0; r < o; ++r)
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
      for (int k = 0; k < D; ++k) t += Hi[r * D + k] * Li[k * D + c];
      HL(r, c) = t;
    }
  for (int r = 0; r < o; ++r)
    for (int s = 0; s <= r; ++s) {
      double t = 0.0;
      for (int c = 0; c < D; ++c) t += HL(r, c) * HL(s, c);
      S(r, s) = t;
      S(s, r) = t;
    }
  chol_lower(S, o);
  // K = L HL' S^-1  (src/filtering.jl:85-86)
  for (int a = 0; a < D; ++a)
    for (int r = 0; r < o; ++r) {
      double t = 0.0;
      for (int c = 0; c < D; ++c) t += Li[a * D + c] * HL(r, c);
      K(a, r) = t;
    }
  solve_right_spd(S, o, K, D);
  for (int a = 0; a < D; ++a) {
    double t = mu[(size_t)i * D + a];
    for (int r = 0; r < o; ++r) t += K(a, r) * (0.0 - z[(size_t)i * o + r]);
    mu_out[(size_t)i * D + a] = t;
  }
  // L <- (I - K H) L = L - K (H L)   (src/filtering.jl:89)
  for (int a = 0; a < D; ++a)
    for (int c = 0; c < D; ++c) {
      double t = Li[a * D + c];
      for (int r = 0; r < o; ++r) t -= K(a, r) * HL(r, c);
      Lf(a, c) = t;
    }
  gram_out(D, Lf, D, cov_out + (size_t)i * D * D);
}

__global__ void smooth_step_kernel(int D, long n, const double* mu, const double* L, const double* mu_s, const double* L_s,
                                   const double* A, const double* QL, double* mu_out, double* cov_out, double* ws) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* Li = L + (size_t)i * D * D;
  const double* Ls = L_s + (size_t)i * D * D;
  const int DD = D * D;
  double* w = ws + (size_t)i * 6 * DD;
  Mat M{w, D}, B{w + DD, D}, G{w + 2 * DD, D}, F{w + 3 * DD, D}, X{w + 4 * DD, 3 * D};  // X: D x 3D stacked factor
  predict_factor(D, A, QL, Li, M, B);  // M = A L, B = chol(P^-)
  // G = Sigma A' (P^-)^-1 = L (A L)' (P^-)^-1   (src/filtering.jl:142)
  for (int a = 0; a < D; ++a)
    for (int b = 0; b < D; ++b) {
      double t = 0.0;
      for (int k = 0; k < D; ++k) t += Li[a * D + k] * M(b, k);
      G(a, b) = t;
    }
  solve_right_spd(B, D, G, D);
  // mean
  for (int a = 0; a < D; ++a) {
    double t = mu[(size_t)i * D + a];
    for (int b = 0; b < D; ++b) {
      double mp = 0.0;
      for (int k = 0; k < D; ++k) mp += A[b * D + k] * mu[(size_t)i * D + k];
      t += G(a, b) * (mu_s[(size_t)i * D + b] - mp);
    }
    mu_out[(size_t)i * D + a] = t;
  }
  // stacked factor [ (I-GA) L | G Q_L | G L_s ]  (transpose of the QR stack, src/filtering.jl:146-148)
  for (int a = 0; a < D; ++a)
    for (int b = 0; b < D; ++b) {
      double t = (a == b) ? 1.0 : 0.0;
      for (int k = 0; k < D; ++k) t -= G(a, k) * A[k * D + b];
      F(a, b) = t;
    }
  for (int a = 0; a < D; ++a)
    for (int c = 0; c < D; ++c) {
      double t0 = 0.0, t1 = 0.0, t2 = 0.0;
      for (int k = 0; k < D; ++k) {
        t0 += F(a, k) * Li[k * D + c];
        t1 += G(a, k) * QL[k * D + c];
        t2 += G(a, k) * Ls[k * D + c];
      }
      X(a, c) = t0;
      X(a, D + c) = t1;
      X(a, 2 * D + c) = t2;
    }
  gram_out(D, X, 3 * D, cov_out + (size_t)i * D * D);
}

thread_local std::string g_err;

struct DevBufs {
  std::vector<void*> ptrs;
  ~DevBufs() { for (void* p : ptrs) (void)hipFree(p); }
  double* up(const double* h, size_t cnt) {
    void* d = nullptr;
    if (hipMalloc(&d, cnt * sizeof(double)) != hipSuccess) return nullptr;
    ptrs.push_back(d);
    if (h && hipMemcpy(d, h, cnt * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return (double*)d;
  }
};

int finish(double* d_mu, double* d_cov, double* mu_out, double* cov_out, int D, int64_t n) {
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return -1;
  if (hipMemcpy(mu_out, d_mu, sizeof(double) * n * D, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (hipMemcpy(cov_out, d_cov, sizeof(double) * n * D * D, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return 0;
}

bool bad_dims(int D, int64_t n) { return D < 1 || D > ODEF_MAX_STEP_DIM || n < 0; }

}  // namespace

extern "C" {

int odef_predict(int D, int64_t n, const double* mu, const double* L, const double* A, const double* Q_L, double* mu_out,
                 double* cov_out) {
  if (bad_dims(D, n) || !mu || !L || !A || !Q_L || !mu_out || !cov_out) return -1;
  if (n == 0) return 0;
  DevBufs b;
  const size_t DD = (size_t)D * D;
  double *dmu = b.up(mu, n * D), *dL = b.up(L, n * DD), *dA = b.up(A, DD), *dQ = b.up(Q_L, DD);
  double *dmo = b.up(nullptr, n * D), *dco = b.up(nullptr, n * DD), *ws = b.up(nullptr, n * 2 * DD);
  if (!dmu || !dL || !dA || !dQ || !dmo || !dco || !ws) return -1;
  hipLaunchKernelGGL(predict_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, D, (long)n, dmu, dL, dA, dQ, dmo, dco, ws);
  return finish(dmo, dco, mu_out, cov_out, D, n);
}

int odef_update(int D, int o, int64_t n, const double* mu_pred, const double* L_pred, const double* H, const double* z,
                double* mu_out, double* cov_out) {
  if (bad_dims(D, n) || o < 1 || o > D || !mu_pred || !L_pred || !H || !z || !mu_out || !cov_out) return -1;
  if (n == 0) return 0;
  DevBufs b;
  const size_t DD = (size_t)D * D;
  double *dmu = b.up(mu_pred, n * D), *dL = b.up(L_pred, n * DD), *dH = b.up(H, n * o * D), *dz = b.up(z, n * o);
  double *dmo = b.up(nullptr, n * D), *dco = b.up(nullptr, n * DD), *ws = b.up(nullptr, n * (2 * o * D + o * o + DD));
  if (!dmu || !dL || !dH || !dz || !dmo || !dco || !ws) return -1;
  hipLaunchKernelGGL(update_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, D, o, (long)n, dmu, dL, dH, dz, dmo, dco, ws);
  return finish(dmo, dco, mu_out, cov_out, D, n);
}

int odef_smooth_step(int D, int64_t n, const double* mu, const double* L, const double* mu_s, const double* L_s,
                     const double* A, const double* Q_L, double* mu_out, double* cov_out) {
  if (bad_dims(D, n) || !mu || !L || !mu_s || !L_s || !A || !Q_L || !mu_out || !cov_out) return -1;
  if (n == 0) return 0;
  DevBufs b;
  const size_t DD = (size_t)D * D;
  double *dmu = b.up(mu, n * D), *dL = b.up(L, n * DD), *dms = b.up(mu_s, n * D), *dLs = b.up(L_s, n * DD);
  double *dA = b.up(A, DD), *dQ = b.up(Q_L, DD);
  double *dmo = b.up(nullptr, n * D), *dco = b.up(nullptr, n * DD), *ws = b.up(nullptr, n * 6 * DD);
  if (!dmu || !dL || !dms || !dLs || !dA || !dQ || !dmo || !dco || !ws) return -1;
  hipLaunchKernelGGL(smooth_step_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, D, (long)n, dmu, dL, dms, dLs, dA, dQ, dmo, dco, ws);
  return finish(dmo, dco, mu_out, cov_out, D, n);
}

}  // extern "C"
