"""CPU oracle for the Gaussian ODE-filter hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a plain numpy/float64 restatement of the arithmetic of the reference
(nathanaelbosch/ODEFilters.jl == ProbNumDiffEq v0.1.5, all citations are `file:line`
relative to the reference tree).  It exists so that the HIP kernels in
`odefilters.jl_amd/csrc/` can be checked for parity.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the
product path (`odefilters.jl_amd`) never does.

Pinning status (see DESIGN.md "Oracle"):
  * the reference is Julia; there is no Julia toolchain in the build image, so the
    reference itself cannot be executed here.  Nothing was denied -- it is absent.
  * pinned by the reference's own tests (restated in tests/test_oracle_kats.py):
      test/priors.jl:25-39,50-59   exact A(h), Q(h) tables (vanilla d=2,q=2; preconditioned d=1,q=2)
      test/preconditioning.jl:32-38  Q_p == P Q(h) P', A_p == P A(h) P^-1, cond(Q(h)) > cond(Q_p)^2
      test/state_init.jl:21-44     Taylor-mode derivatives a^k u0 up to q=6
      test/filtering.jl:28-46,79-89,118-122   predict / update / smooth identities
      test/solution.jl:38-41       mu_0 == u0, Sigma_0 == 0
      test/correctness.jl:33-35    fixed-step rtol 1e-5 vs a 1e-20 solution (here: DOP853 @1e-13 / mpmath)
      test/correctness.jl:62-66    adaptive rtol 1e-3
      test/convergence.jl:17-38    order q+1
      test/smoothing.jl:31-44      smoothed error < 2 x filter error, last state equal
  * PARITY UNPINNED (third-party arithmetic with no reference test on it):
      `sol.log_likelihood` (GaussianDistributions 0.5 `logpdf`, perform_step.jl:66),
      the adaptive step *sequence* (OrdinaryDiffEq 5 PI controller / fastpow / initdt;
      restated from its published algorithm, see `solve`), and
      test/specific_problems.jl:147-155 (ForwardDiff gradients; needs Julia AD).

Conventions: state ordering is derivative-major x = [u; u'; ...; u^(q)] (caches.jl:63-64),
a Gaussian is (mu[D], L[D,D]) with Sigma = L L^T (`SquarerootMatrix`, squarerootmatrix.jl:10-16).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

# --------------------------------------------------------------------------------------
# L1: square-root covariance type
# --------------------------------------------------------------------------------------


@dataclass
class SRGaussian:
    """`Gaussian{Vector, SRMatrix}` (ProbNumDiffEq.jl:47).  `L` is `Σ.squareroot`,
    `cov()` is `Σ.mat = S*S'` (squarerootmatrix.jl:16)."""

    mu: np.ndarray
    L: np.ndarray

    def cov(self) -> np.ndarray:
        return self.L @ self.L.T

    def copy(self) -> "SRGaussian":
        return SRGaussian(self.mu.copy(), self.L.copy())


def X_A_Xt_sr(L: np.ndarray, X: np.ndarray) -> np.ndarray:
    """squarerootmatrix.jl:38-39: X_A_Xt(M::SRMatrix, X) = SRMatrix(X*M.squareroot)."""
    return X @ L


def linmap(M_diag: np.ndarray, g: SRGaussian) -> SRGaussian:
    """ProbNumDiffEq.jl:58  `M * g` for a Diagonal M (the preconditioner): row scaling."""
    return SRGaussian(M_diag * g.mu, M_diag[:, None] * g.L)


def apply_diffusion(Q_L: np.ndarray, diffusion: float) -> np.ndarray:
    """ProbNumDiffEq.jl:39: SRMatrix(sqrt(diffusion) * Q.squareroot)."""
    return math.sqrt(diffusion) * Q_L


# --------------------------------------------------------------------------------------
# L0: model constants
# --------------------------------------------------------------------------------------


def ibm(d: int, q: int):
    """priors.jl:7-59.  Preconditioned, h-independent IBM transition `A` (upper
    triangular, A[j, j+d*i] = 1/i!) and the lower Cholesky factor `Q_L` of
    Q[r,c] = 1/((2q+1-r-c)(q-r)!(q-c)!) (x) I_d."""
    D = d * (q + 1)
    A = np.eye(D)
    val = 1.0
    for i in range(1, q + 1):  # priors.jl:17-22
        val = val / i
        for j in range(d * (q + 1 - i)):
            A[j, j + d * i] = val
    Q = np.zeros((D, D))
    for col in range(q + 1):  # priors.jl:41-51
        for row in range(col, q + 1):
            idx = 2 * q + 1 - row - col
            v = 1.0 / (idx * math.factorial(q - row) * math.factorial(q - col))
            for i in range(d):
                Q[col * d + i, row * d + i] = v
                Q[row * d + i, col * d + i] = v
    Q_L = np.linalg.cholesky(Q)  # priors.jl:54
    return A, Q_L


def vanilla_ibm(d: int, q: int):
    """priors.jl:63-98 (test-only in the reference): un-preconditioned A(h), Q(h, sigma^2)."""
    D = d * (q + 1)

    def A_of(h):
        A = np.eye(D)
        val = 1.0
        for i in range(1, q + 1):
            val = val * h / i
            for j in range(d * (q + 1 - i)):
                A[j, j + d * i] = val
        return A

    def Q_of(h, sigma2=1.0):
        Q = np.zeros((D, D))
        for col in range(q + 1):
            for row in range(col, q + 1):
                idx = 2 * q + 1 - row - col
                v = h**idx / (idx * math.factorial(q - row) * math.factorial(q - col)) * sigma2
                for i in range(d):
                    Q[col * d + i, row * d + i] = v
                    Q[row * d + i, col * d + i] = v
        return Q

    return A_of, Q_of


def preconditioner(d: int, q: int) -> Callable[[float], np.ndarray]:
    """preconditioning.jl:1-17.  P(h) = diag(h^(j-q-1/2)) repeated d times, built as the
    reference does: val = h^(-q-1/2), then `val *= h` per derivative (same rounding)."""

    def P(h: float) -> np.ndarray:
        out = np.empty(d * (q + 1))
        val = h ** (-q - 1 / 2)
        for j in range(q + 1):
            out[j * d : (j + 1) * d] = val
            val *= h
        return out

    return P


def projection(d: int, q: int, deriv: int) -> np.ndarray:
    """caches.jl:63-64  Proj(deriv) = e_{deriv+1}' (x) I_d."""
    if deriv > q:
        raise ValueError("Projection called for non-modeled derivative")
    E = np.zeros((d, d * (q + 1)))
    E[:, deriv * d : (deriv + 1) * d] = np.eye(d)
    return E


# --------------------------------------------------------------------------------------
# L2: Kalman algebra (filtering.jl)
# --------------------------------------------------------------------------------------


def predict_mean(mu: np.ndarray, A: np.ndarray) -> np.ndarray:
    """filtering.jl:22-25."""
    return A @ mu


def predict_cov_sr(L: np.ndarray, A: np.ndarray, Qh_L: np.ndarray):
    """filtering.jl:33-48.  Returns (L_pred, used_qr).  Gram + Cholesky; when the
    Cholesky fails (`issuccess` false) fall back to the QR of the stacked factor."""
    _L = np.hstack([A @ L, Qh_L])
    out_cov = _L @ _L.T
    try:
        if not np.all(np.isfinite(out_cov)):
            raise np.linalg.LinAlgError
        PpL = np.linalg.cholesky(out_cov)
        return PpL, False
    except np.linalg.LinAlgError:
        R = np.linalg.qr(_L.T, mode="r")
        return np.tril(R.T), True


def predict(x: SRGaussian, A: np.ndarray, Qh_L: np.ndarray) -> SRGaussian:
    """filtering.jl:17-21,56-60."""
    Lp, _ = predict_cov_sr(x.L, A, Qh_L)
    return SRGaussian(predict_mean(x.mu, A), Lp)


def update(x_pred: SRGaussian, z: np.ndarray, S: np.ndarray, H: np.ndarray) -> SRGaussian:
    """filtering.jl:79-91 with R == 0:  K = P_p H' S^-1;  m = m_p + K(0 - z);
    Sigma <- X_A_Xt(P_p, I - K H) i.e. factor (I - K H) L_p."""
    P_p = x_pred.L @ x_pred.L.T  # SRMatrix.mat (squarerootmatrix.jl:16), read by `P_p * H'`
    S_inv = np.linalg.inv(S)
    K = P_p @ H.T @ S_inv
    m = x_pred.mu + K @ (0.0 - z)
    D = len(m)
    L = (np.eye(D) - K @ H) @ x_pred.L
    return SRGaussian(m, L)


def smooth(x_curr: SRGaussian, x_next_smoothed: SRGaussian, A: np.ndarray, Qh_L: np.ndarray):
    """filtering.jl:136-154 and the in-place twin smoothing.jl:31-63 (same arithmetic).
    Returns (x_smoothed, G)."""
    x_pred = predict(x_curr, A, Qh_L)
    P_p = x_pred.L @ x_pred.L.T
    P_p_inv = np.linalg.inv(P_p)  # squarerootmatrix.jl:42
    Sigma = x_curr.L @ x_curr.L.T
    G = Sigma @ A.T @ P_p_inv
    m = x_curr.mu + G @ (x_next_smoothed.mu - x_pred.mu)
    D = len(m)
    _R = np.vstack(
        [
            x_curr.L.T @ (np.eye(D) - G @ A).T,
            Qh_L.T @ G.T,
            x_next_smoothed.L.T @ G.T,
        ]
    )
    P_s_R = np.linalg.qr(_R, mode="r")
    return SRGaussian(m, P_s_R.T), G


def smooth_dense(m, P, m_s, P_s, A, Q):
    """Textbook RTS used by test/filtering.jl:111-113 (covariance form)."""
    m_p = A @ m
    P_p = A @ P @ A.T + Q
    G = P @ A.T @ np.linalg.inv(P_p)
    return m + G @ (m_s - m_p), P + G @ (P_s - P_p) @ G.T


# --------------------------------------------------------------------------------------
# Vector fields (the build's RHS registry; ids must match include/odefilter.h)
# --------------------------------------------------------------------------------------


class Jet:
    """Truncated univariate Taylor arithmetic, coefficients c[k] of (t-t0)^k.
    Used for the Taylor-mode initialisation (state_initialization.jl:15-42 computes the
    same exact derivatives with TaylorSeries.jl's multivariate Lie recursion)."""

    __slots__ = ("c",)

    def __init__(self, c):
        self.c = np.asarray(c, dtype=float)

    @staticmethod
    def lift(x, n):
        if isinstance(x, Jet):
            return x
        c = np.zeros(n)
        c[0] = x
        return Jet(c)

    def __add__(self, o):
        o = Jet.lift(o, len(self.c))
        return Jet(self.c + o.c)

    __radd__ = __add__

    def __neg__(self):
        return Jet(-self.c)

    def __sub__(self, o):
        o = Jet.lift(o, len(self.c))
        return Jet(self.c - o.c)

    def __rsub__(self, o):
        return Jet.lift(o, len(self.c)) - self

    def __mul__(self, o):
        if not isinstance(o, Jet):
            return Jet(self.c * o)
        n = len(self.c)
        out = np.zeros(n)
        for k in range(n):
            out[k] = np.dot(self.c[: k + 1], o.c[k::-1])
        return Jet(out)

    __rmul__ = __mul__

    def __truediv__(self, o):
        if not isinstance(o, Jet):
            return Jet(self.c / o)
        n = len(self.c)
        out = np.zeros(n)
        for k in range(n):
            out[k] = (self.c[k] - np.dot(out[:k], o.c[k:0:-1])) / o.c[0]
        return Jet(out)

    def __rtruediv__(self, o):
        return Jet.lift(o, len(self.c)) / self

    def __pow__(self, a):
        if isinstance(a, int) and a >= 0:
            out = Jet.lift(1.0, len(self.c))
            for _ in range(a):
                out = out * self
            return out
        # real power: k x0 p_k = sum_{j=1..k} (a j - (k - j)) x_j p_{k-j}
        n = len(self.c)
        out = np.zeros(n)
        out[0] = self.c[0] ** a
        for k in range(1, n):
            s = 0.0
            for j in range(1, k + 1):
                s += (a * j - (k - j)) * self.c[j] * out[k - j]
            out[k] = s / (k * self.c[0])
        return Jet(out)


@dataclass
class VectorField:
    name: str
    rhs_id: int
    d: int
    n_params: int
    f: Callable  # f(u, p, t) -> list of d scalars (generic in the scalar type)
    jac: Callable  # jac(u, p, t) -> np.ndarray d x d
    u0: np.ndarray
    p: np.ndarray
    tspan: tuple


def _fhn_f(u, p, t):
    a, b, c = p
    return [c * (u[0] - u[0] * u[0] * u[0] / 3.0 + u[1]), -(1.0 / c) * (u[0] - a - b * u[1])]


def _fhn_jac(u, p, t):
    a, b, c = p
    return np.array([[c * (1.0 - u[0] * u[0]), c], [-(1.0 / c), b / c]])


def _lorenz_f(u, p, t):
    s, r, b = p
    return [s * (u[1] - u[0]), u[0] * (r - u[2]) - u[1], u[0] * u[1] - b * u[2]]


def _lorenz_jac(u, p, t):
    s, r, b = p
    return np.array([[-s, s, 0.0], [r - u[2], -1.0, -u[0]], [u[1], u[0], -b]])


def _lv_f(u, p, t):
    a, b, c, dd = p
    return [a * u[0] - b * u[0] * u[1], -c * u[1] + dd * u[0] * u[1]]


def _lv_jac(u, p, t):
    a, b, c, dd = p
    return np.array([[a - b * u[1], -b * u[0]], [dd * u[1], -c + dd * u[0]]])


def _vdp_f(u, p, t):
    (mu,) = p
    return [u[1], mu * ((1.0 - u[0] * u[0]) * u[1] - u[0])]


def _vdp_jac(u, p, t):
    (mu,) = p
    return np.array([[0.0, 1.0], [mu * (-2.0 * u[0] * u[1] - 1.0), mu * (1.0 - u[0] * u[0])]])


def _lin_f(u, p, t):
    return [p[i] * u[i] for i in range(len(u))]


def _lin_jac(u, p, t):
    return np.diag(np.asarray(p, dtype=float)[: len(u)])


def _pleiades_f(u, p, t):
    # state (x1..7, y1..7, vx1..7, vy1..7), masses m_i = i (Hairer IVP test set)
    x, y, vx, vy = u[0:7], u[7:14], u[14:21], u[21:28]
    ax, ay = [], []
    for i in range(7):
        sx, sy = 0.0, 0.0
        for j in range(7):
            if j == i:
                continue
            dx = x[j] - x[i]
            dy = y[j] - y[i]
            r2 = dx * dx + dy * dy
            w = (j + 1.0) * (r2 ** (-1.5))
            sx = sx + w * dx
            sy = sy + w * dy
        ax.append(sx)
        ay.append(sy)
    return list(vx) + list(vy) + ax + ay


def _pleiades_jac(u, p, t):
    x, y = np.asarray(u[0:7], float), np.asarray(u[7:14], float)
    J = np.zeros((28, 28))
    J[0:7, 14:21] = np.eye(7)
    J[7:14, 21:28] = np.eye(7)
    for i in range(7):
        for j in range(7):
            if j == i:
                continue
            mj = j + 1.0
            dx, dy = x[j] - x[i], y[j] - y[i]
            r2 = dx * dx + dy * dy
            r3 = r2**-1.5
            r5 = r2**-2.5
            # d(ax_i)/d(x_j) etc
            axx = mj * (r3 - 3.0 * dx * dx * r5)
            axy = mj * (-3.0 * dx * dy * r5)
            ayy = mj * (r3 - 3.0 * dy * dy * r5)
            J[14 + i, j] += axx
            J[14 + i, i] -= axx
            J[14 + i, 7 + j] += axy
            J[14 + i, 7 + i] -= axy
            J[21 + i, j] += axy
            J[21 + i, i] -= axy
            J[21 + i, 7 + j] += ayy
            J[21 + i, 7 + i] -= ayy
    return J


def _l96_f(u, p, t):
    n = len(u)
    return [(u[(i + 1) % n] - u[(i - 2) % n]) * u[(i - 1) % n] - u[i] + p[0] for i in range(n)]


def _l96_jac(u, p, t):
    n = len(u)
    J = np.zeros((n, n))
    for i in range(n):
        ip, im2, im1 = (i + 1) % n, (i - 2) % n, (i - 1) % n
        J[i, ip] += u[im1]
        J[i, im2] -= u[im1]
        J[i, im1] += u[ip] - u[im2]
        J[i, i] -= 1.0
    return J


RHS_FHN, RHS_LORENZ63, RHS_LOTKA_VOLTERRA, RHS_VANDERPOL, RHS_LINEAR, RHS_PLEIADES, RHS_LORENZ96 = 0, 1, 2, 3, 4, 5, 6


def vector_field(name: str) -> VectorField:
    if name == "fhn":  # examples/fitzhughnagumo_animation.jl:8-16, README.md:36-44
        return VectorField("fhn", RHS_FHN, 2, 3, _fhn_f, _fhn_jac, np.array([-1.0, 1.0]), np.array([0.2, 0.2, 3.0]), (0.0, 20.0))
    if name == "lorenz63":
        return VectorField("lorenz63", RHS_LORENZ63, 3, 3, _lorenz_f, _lorenz_jac, np.array([1.0, 0.0, 0.0]), np.array([10.0, 28.0, 8.0 / 3.0]), (0.0, 2.0))
    if name == "lotka_volterra":  # DiffEqProblemLibrary prob_ode_lotkavoltera (third-party, recalled)
        return VectorField("lotka_volterra", RHS_LOTKA_VOLTERRA, 2, 4, _lv_f, _lv_jac, np.array([1.0, 1.0]), np.array([1.5, 1.0, 3.0, 1.0]), (0.0, 1.0))
    if name == "vanderpol":  # test/specific_problems.jl:44-47 (stiff, mu = 1e6 there)
        return VectorField("vanderpol", RHS_VANDERPOL, 2, 1, _vdp_f, _vdp_jac, np.array([2.0, 0.0]), np.array([1.0]), (0.0, 6.3))
    if name == "linear":  # test/convergence.jl:9-14, test/state_init.jl:12-17
        return VectorField("linear", RHS_LINEAR, 2, 2, _lin_f, _lin_jac, np.array([0.1, 1.0]), np.array([1.1, -0.5]), (0.0, 5.0))
    if name == "pleiades":
        u0 = np.array(
            [3.0, 3.0, -1.0, -3.0, 2.0, -2.0, 2.0, 3.0, -3.0, 2.0, 0.0, 0.0, -4.0, 4.0]
            + [0.0, 0.0, 0.0, 0.0, 0.0, 1.75, -1.5, 0.0, 0.0, 0.0, -1.25, 1.0, 0.0, 0.0]
        )
        return VectorField("pleiades", RHS_PLEIADES, 28, 0, _pleiades_f, _pleiades_jac, u0, np.zeros(0), (0.0, 0.25))
    if name == "lorenz96":  # 16 variables: a second shape for the workgroup-per-trajectory kernels (no counterpart in the reference's tests)
        u0 = 8.0 + np.array([0.01 * ((7 * i) % 5 - 2) for i in range(16)])
        u0[0] += 1.0
        return VectorField("lorenz96", RHS_LORENZ96, 16, 1, _l96_f, _l96_jac, u0, np.array([8.0]), (0.0, 0.25))
    raise KeyError(name)


# --------------------------------------------------------------------------------------
# State initialisation (state_initialization.jl)
# --------------------------------------------------------------------------------------


def get_derivatives(u0: np.ndarray, vf: VectorField, p, t0: float, q: int) -> List[np.ndarray]:
    """state_initialization.jl:15-42: exact u'(t0)..u^(q)(t0) of the autonomous ODE.
    Restated as the classical Taylor-coefficient recursion u_{k+1} = f(u)_k/(k+1)
    (the reference gets the same derivatives from a multivariate Lie recursion)."""
    d = len(u0)
    n = q + 1
    coef = np.zeros((d, n))
    coef[:, 0] = u0
    for k in range(q):
        jets = [Jet(coef[i].copy()) for i in range(d)]
        fu = vf.f(jets, p, t0)
        for i in range(d):
            fi = Jet.lift(fu[i], n)
            coef[i, k + 1] = fi.c[k] / (k + 1)
    return [coef[:, k] * math.factorial(k) for k in range(1, q + 1)]


def condition_on(x: SRGaussian, H: np.ndarray, data: np.ndarray) -> SRGaussian:
    """state_initialization.jl:45-53."""
    z = H @ x.mu
    Sig = x.L @ x.L.T
    S = (H @ x.L) @ (H @ x.L).T
    K = Sig @ H.T @ np.linalg.inv(S)
    mu = x.mu + K @ (data - z)
    L = (np.eye(len(mu)) - K @ H) @ x.L
    return SRGaussian(mu, L)


def initial_update(u0: np.ndarray, vf: VectorField, p, t0: float, q: int) -> SRGaussian:
    """state_initialization.jl:2-14 applied to caches.jl:73  x0 = N(0, I)."""
    d = len(u0)
    D = d * (q + 1)
    x = SRGaussian(np.zeros(D), np.eye(D))
    x = condition_on(x, projection(d, q, 0), np.asarray(u0, float))
    for o, df in zip(range(1, q + 1), get_derivatives(u0, vf, p, t0, q)):
        x = condition_on(x, projection(d, q, o), df)
    return x


# --------------------------------------------------------------------------------------
# L3: step orchestration (perform_step.jl) + diffusions.jl
# --------------------------------------------------------------------------------------


@dataclass
class Alg:
    """algorithms.jl:23-28,46-51 (EK0 / EK1 keyword structs)."""

    kind: str = "EK1"  # "EK0" | "EK1"
    order: int = 3
    diffusionmodel: str = "dynamic"  # "dynamic" | "fixed" | "fixedMAP"
    smooth: bool = True


def EK0(order=3, diffusionmodel="dynamic", smooth=True):
    return Alg("EK0", order, diffusionmodel, smooth)


def EK1(order=3, diffusionmodel="dynamic", smooth=True):
    return Alg("EK1", order, diffusionmodel, smooth)


@dataclass
class StepResult:
    x_filt: SRGaussian
    x_pred: SRGaussian
    x_back: SRGaussian  # PI * (P * x): what cache.x holds after a rejected step (perform_step.jl:73)
    u_filt: np.ndarray
    local_diffusion: float
    global_diffusion: float
    log_likelihood: float
    H: np.ndarray
    z: np.ndarray
    S: np.ndarray
    used_qr: bool


def logpdf_zero(z: np.ndarray, S: np.ndarray) -> float:
    """perform_step.jl:66 `logpdf(measurement, zeros(d))` (GaussianDistributions 0.5,
    third-party; published formula -(z'S^-1 z + logdet S + d log 2pi)/2).  Unpinned."""
    d = len(z)
    try:
        c = np.linalg.cholesky(S)
    except np.linalg.LinAlgError:
        return float("nan")
    w = np.linalg.solve(c, z)
    return -0.5 * (float(w @ w) + 2.0 * float(np.sum(np.log(np.diag(c)))) + d * math.log(2.0 * math.pi))


def measure(alg: Alg, vf: VectorField, p, m_pred: np.ndarray, PI: np.ndarray, t: float, d: int, q: int):
    """perform_step.jl:95-132 (mean part and H; the `S` computed at :129 in the dynamic
    branch is overwritten at :54 and not reproduced)."""
    E0, E1 = projection(d, q, 0), projection(d, q, 1)
    u_pred = E0 @ (PI * m_pred)
    du = np.asarray(vf.f(list(u_pred), p, t), float)
    z = E1 @ (PI * m_pred) - du
    if alg.kind == "EK1":
        ddu = vf.jac(u_pred, p, t)
        H = (E1 - ddu @ E0) * PI[None, :]
    else:
        H = E1 * PI[None, :]
    return z, H, u_pred


def estimate_diffusion_dynamic(z: np.ndarray, H: np.ndarray, Q_L: np.ndarray, d: int) -> float:
    """diffusions.jl:72-80:  sigma^2 = z' ((H Q H') \\ z) / d."""
    HQ = H @ Q_L
    return float(z @ np.linalg.solve(HQ @ HQ.T, z)) / d


def estimate_errors(local_diffusion: float, Q_L: np.ndarray, H: np.ndarray):
    """perform_step.jl:148-158."""
    if math.isinf(local_diffusion):
        return np.full(H.shape[0], np.inf)
    HQ = H @ apply_diffusion(Q_L, local_diffusion)
    return np.sqrt(np.diag(HQ @ HQ.T))


def perform_step(alg: Alg, vf: VectorField, p, consts, x: SRGaussian, t: float, dt: float,
                 success_iter: int = 0, prev_global_diffusion: Optional[float] = None) -> StepResult:
    """perform_step.jl:27-76 (everything up to the error estimate)."""
    A, Q_L, precond, d, q = consts
    tnew = t + dt
    P = precond(dt)
    PI = 1.0 / P  # inv(::Diagonal)
    xp = linmap(P, x)  # :38

    used_qr = False
    if alg.diffusionmodel == "dynamic":  # :40-54
        m_pred = predict_mean(xp.mu, A)
        z, H, _ = measure(alg, vf, p, m_pred, PI, tnew, d, q)
        sigma2 = estimate_diffusion_dynamic(z, H, Q_L, d)
        local_diffusion = global_diffusion = sigma2
        L_pred, used_qr = predict_cov_sr(xp.L, A, apply_diffusion(Q_L, global_diffusion))
        HL = H @ L_pred
        S = HL @ HL.T
    elif alg.diffusionmodel in ("fixed", "fixedMAP"):  # :56-63, diffusions.jl:11-36 / :46-68
        m_pred = predict_mean(xp.mu, A)
        L_pred, used_qr = predict_cov_sr(xp.L, A, Q_L)
        z, H, _ = measure(alg, vf, p, m_pred, PI, tnew, d, q)
        HL = H @ L_pred
        S = HL @ HL.T
        diffusion_t = float(z @ np.linalg.inv(S) @ z) / d
        local_diffusion = diffusion_t
        if alg.diffusionmodel == "fixed":
            if success_iter == 0:
                global_diffusion = diffusion_t
            else:
                global_diffusion = prev_global_diffusion + (diffusion_t - prev_global_diffusion) / success_iter
        else:  # MAPFixedDiffusion (diffusions.jl:46-68): mode of the InverseGamma(1/2, 1/2) posterior, on-line
            n_obs = success_iter + 1
            alpha, beta = 1 / 2, 1 / 2
            if success_iter == 0:
                global_diffusion = (beta + 1 / 2 * diffusion_t) / (alpha + n_obs * d / 2 + 1)
            else:
                res_prev = (prev_global_diffusion * (alpha + (n_obs - 1) * d / 2 + 1) - beta) * 2
                res_sum_t = res_prev + diffusion_t
                global_diffusion = (beta + 1 / 2 * res_sum_t) / (alpha + n_obs * d / 2 + 1)
    else:
        raise NotImplementedError(alg.diffusionmodel)
    x_pred = SRGaussian(m_pred, L_pred)

    ll = logpdf_zero(z, S)  # :66
    x_filt = update(x_pred, z, S, H)  # :69
    u_filt = (PI * x_filt.mu)[:d]  # :70
    return StepResult(
        x_filt=linmap(PI, x_filt),  # :75
        x_pred=linmap(PI, x_pred),  # :74
        x_back=linmap(PI, xp),  # :73
        u_filt=u_filt,
        local_diffusion=local_diffusion,
        global_diffusion=global_diffusion,
        log_likelihood=ll,
        H=H,
        z=z,
        S=S,
        used_qr=used_qr,
    )


def internalnorm(u: np.ndarray) -> float:
    """DiffEqBase ODE_DEFAULT_NORM for arrays: sqrt(sum(abs2,u)/length(u)) (third-party)."""
    return math.sqrt(float(np.sum(u * u)) / len(u))


def calculate_EEst(alg_consts, res: StepResult, dt: float, u_prev: np.ndarray, abstol: float, reltol: float) -> float:
    """perform_step.jl:78-84 + DiffEqBase.calculate_residuals! (third-party):
    err_i = dt*e_i / (abstol + max(|u_i|,|u_filt_i|)*reltol); EEst = RMS(err)."""
    _, Q_L, _, d, q = alg_consts
    e = estimate_errors(res.local_diffusion, Q_L, res.H)
    err = dt * e / (abstol + np.maximum(np.abs(u_prev), np.abs(res.u_filt)) * reltol)
    return internalnorm(err)


# --------------------------------------------------------------------------------------
# Solution container + integrator loop (OrdinaryDiffEq.solve!, third-party, restated)
# --------------------------------------------------------------------------------------


@dataclass
class Solution:
    """solution.jl:8-24 fields that the hot path produces."""

    t: List[float] = field(default_factory=list)
    x_filt: List[SRGaussian] = field(default_factory=list)
    x_smooth: Optional[List[SRGaussian]] = None
    diffusions: List[float] = field(default_factory=list)
    log_likelihood: float = 0.0
    naccept: int = 0
    nreject: int = 0
    nf: int = 0
    njacs: int = 0
    retcode: str = "Success"
    d: int = 0
    q: int = 0
    used_qr: int = 0

    def means(self, smoothed=None) -> np.ndarray:
        xs = self.x_smooth if (smoothed or (smoothed is None and self.x_smooth is not None)) else self.x_filt
        return np.array([x.mu for x in xs])

    def covs(self, smoothed=None) -> np.ndarray:
        xs = self.x_smooth if (smoothed or (smoothed is None and self.x_smooth is not None)) else self.x_filt
        return np.array([x.cov() for x in xs])

    @property
    def u(self) -> np.ndarray:
        """sol.u: E0 * mean (smoothed when available, integrator_utils.jl:20-26)."""
        return self.means()[:, : self.d]


@dataclass
class Controller:
    """OrdinaryDiffEq 5 PI controller defaults with the reference's exponents
    (alg_utils.jl:23-24: beta2 = 2/(5(q+1)), beta1 = 7/(10(q+1)))."""

    beta1: float
    beta2: float
    gamma: float = 0.9
    qmin: float = 0.2
    qmax: float = 10.0
    qsteady_min: float = 1.0
    qsteady_max: float = 1.0
    qoldinit: float = 1e-4

    @staticmethod
    def default(q: int) -> "Controller":
        return Controller(beta1=7.0 / (10.0 * (q + 1)), beta2=2.0 / (5.0 * (q + 1)))


def make_consts(d: int, q: int):
    A, Q_L = ibm(d, q)
    return (A, Q_L, preconditioner(d, q), d, q)


def fixed_time_grid(t0: float, t1: float, dt: float) -> np.ndarray:
    """OrdinaryDiffEq fixed-step grid (third-party, restated): t += dt with the last
    step clipped to the tstop and a 100-eps snap onto it."""
    ts = [t0]
    t = t0
    while t < t1:
        h = min(dt, t1 - t)
        tn = t + h
        if abs(tn - t1) < 100 * np.finfo(float).eps * max(abs(tn), abs(t1)):
            tn = t1
        ts.append(tn)
        t = tn
    return np.array(ts)


def solve(vf: VectorField, alg: Alg, *, u0=None, p=None, tspan=None, dt: Optional[float] = None,
          adaptive: bool = False, abstol: float = 1e-6, reltol: float = 1e-3,
          controller: Optional[Controller] = None, maxiters: int = 100000,
          tgrid: Optional[Sequence[float]] = None) -> Solution:
    """`solve(prob, alg; adaptive, dt, abstol, reltol)` for one trajectory.
    Loop = OrdinaryDiffEq.solve! (third-party, restated): perform_step! ->
    PI controller accept/reject -> savevalues! (integrator_utils.jl:33-48) ->
    postamble! (integrator_utils.jl:2-30)."""
    u0 = vf.u0 if u0 is None else np.asarray(u0, float)
    p = vf.p if p is None else np.asarray(p, float)
    tspan = vf.tspan if tspan is None else tspan
    d, q = len(u0), alg.order
    consts = make_consts(d, q)
    t0, t1 = tspan
    if not adaptive and dt is None and tgrid is None:
        raise ValueError("Fixed timestep methods require a choice of dt or choosing the tstops")  # test/errors.jl:17-19

    x = initial_update(u0, vf, p, t0, q)  # perform_step.jl:2-12
    sol = Solution(d=d, q=q)
    sol.t.append(t0)
    sol.x_filt.append(x.copy())
    u_cur = np.asarray(u0, float).copy()  # integ.u

    if not adaptive:
        grid = fixed_time_grid(t0, t1, dt) if tgrid is None else np.asarray(tgrid, float)
        for n in range(len(grid) - 1):
            t, h = grid[n], grid[n + 1] - grid[n]
            prev_gd = sol.diffusions[-1] if sol.diffusions else None
            res = perform_step(alg, vf, p, consts, x, t, h, success_iter=sol.naccept, prev_global_diffusion=prev_gd)
            sol.nf += 1
            sol.njacs += 1 if alg.kind == "EK1" else 0
            sol.used_qr += int(res.used_qr)
            x = res.x_filt  # perform_step.jl:89-92
            u_cur = res.u_filt
            sol.log_likelihood += res.log_likelihood
            sol.naccept += 1
            sol.t.append(grid[n + 1])
            sol.x_filt.append(x.copy())
            sol.diffusions.append(res.global_diffusion)
            if not np.all(np.isfinite(x.mu)):
                sol.retcode = "Unstable"
                break
    else:
        ctrl = controller or Controller.default(q)
        t = t0
        h = dt if dt is not None else 1e-3
        qold = ctrl.qoldinit
        q11 = 1.0
        iters = 0
        while t < t1:
            iters += 1
            if iters > maxiters:
                sol.retcode = "MaxIters"
                break
            h = min(h, t1 - t)  # tstop clipping
            prev_gd = sol.diffusions[-1] if sol.diffusions else None
            res = perform_step(alg, vf, p, consts, x, t, h, success_iter=sol.naccept, prev_global_diffusion=prev_gd)
            sol.nf += 1
            sol.njacs += 1 if alg.kind == "EK1" else 0
            sol.used_qr += int(res.used_qr)
            EEst = calculate_EEst(consts, res, h, u_cur, abstol, reltol)
            u_cur = res.u_filt  # perform_step.jl:86 (also on rejection)
            if not math.isfinite(EEst):
                EEst = float("inf")
            # stepsize_controller! (PI)
            if EEst == 0.0:
                qq = 1.0 / ctrl.qmax
            else:
                q11 = EEst**ctrl.beta1
                qq = q11 / (qold**ctrl.beta2)
                qq = max(1.0 / ctrl.qmax, min(1.0 / ctrl.qmin, qq / ctrl.gamma))
            if EEst <= 1.0:  # accept (OrdinaryDiffEq accepts on <=)
                if EEst < 1.0:  # perform_step.jl:89: commit only on strict <
                    x = res.x_filt
                    sol.log_likelihood += res.log_likelihood
                else:
                    x = res.x_back
                if ctrl.qsteady_min <= qq <= ctrl.qsteady_max:
                    qq = 1.0
                qold = max(EEst, ctrl.qoldinit)
                tn = t + h
                if abs(tn - t1) < 100 * np.finfo(float).eps * max(abs(tn), abs(t1)):
                    tn = t1
                t = tn
                sol.naccept += 1
                sol.t.append(t)
                sol.x_filt.append(x.copy())
                sol.diffusions.append(res.global_diffusion)
                h = h / qq
            else:
                x = res.x_back  # perform_step.jl:73 leaves cache.x = P^-1 (P x)
                sol.nreject += 1
                h = h / min(1.0 / ctrl.qmin, q11 / ctrl.gamma)
            if not np.all(np.isfinite(x.mu)):
                sol.retcode = "Unstable"
                break

    # postamble! (integrator_utils.jl:2-30)
    if alg.diffusionmodel in ("fixed", "fixedMAP") and sol.diffusions:  # isstatic
        final = sol.diffusions[-1]
        sol.log_likelihood = float("nan")
        for s in sol.x_filt:
            s.L = math.sqrt(final) * s.L
        sol.diffusions = [final for _ in sol.diffusions]
    if alg.smooth:
        smooth_all(sol, consts)
    return sol


def smooth_all(sol: Solution, consts) -> None:
    """smoothing.jl:4-28.  Index 1 (Julia) is never smoothed (loop runs to 2)."""
    A, Q_L, precond, d, q = consts
    x = [g.copy() for g in sol.x_filt]
    t = sol.t
    n = len(x)
    for i in range(n - 2, 0, -1):  # Julia i = N-1 .. 2  ->  python i = N-2 .. 1
        h = t[i + 1] - t[i]
        if h == 0:
            x[i] = x[i + 1].copy()
            continue
        P = precond(h)
        PI = 1.0 / P
        Qh = apply_diffusion(Q_L, sol.diffusions[i])  # Julia diffusions[i] (1-based) == step t[i]->t[i+1]
        xs, _ = smooth(linmap(P, x[i]), linmap(P, x[i + 1]), A, Qh)
        if np.any(np.isnan(xs.mu)) or np.any(np.isnan(xs.L)):
            raise AssertionError("NaNs after smoothing")  # smoothing.jl:25
        x[i] = linmap(PI, xs)
    sol.x_smooth = x


def dense_output(sol: Solution, consts, tval: float, smoothed: bool = True) -> SRGaussian:
    """solution.jl:165-210: posterior at an arbitrary time."""
    A, Q_L, precond, d, q = consts
    t = np.asarray(sol.t)
    if tval < t[0]:
        raise ValueError("Invalid t<t0")
    idx = int(np.sum(t <= tval))  # 1-based like Julia
    if np.any(t == tval):
        return (sol.x_smooth if smoothed else sol.x_filt)[idx - 1]
    prev_rv = sol.x_filt[idx - 1]
    diffusion = sol.diffusions[min(idx, len(sol.diffusions)) - 1]
    h1 = tval - t[idx - 1]
    P = precond(h1)
    Qh = apply_diffusion(Q_L, diffusion)
    goal_pred = linmap(1.0 / P, predict(linmap(P, prev_rv), A, Qh))
    if not smoothed or tval >= t[-1]:
        return goal_pred
    h2 = t[idx] - tval
    P = precond(h2)
    gs, _ = smooth(linmap(P, goal_pred), linmap(P, sol.x_smooth[idx]), A, Qh)
    return linmap(1.0 / P, gs)


# --------------------------------------------------------------------------------------
# Posterior sampling (solution_sampling.jl:24-62)
# --------------------------------------------------------------------------------------


def sample_normal(seed: int, traj: int, sample: int, slot: int, k: int, n_samples: int, n_save: int, D: int) -> float:
    """The build's reproducible N(0,1) stream (the reference draws from Julia's global RNG, which cannot be
    reproduced; only the distribution is part of the contract).  Counter c = ((traj*n_samples + sample)*n_save
    + slot)*D + k;  U1, U2 = (splitmix64(seed + 2c [+1]) >> 11) * 2^-53;  Box-Muller on (1 - U1, U2)."""
    c = (((traj * n_samples + sample) * n_save + slot) * D + k) & _M64
    u1 = (splitmix64((seed + 2 * c) & _M64) >> 11) * 2.0**-53
    u2 = (splitmix64((seed + 2 * c + 1) & _M64) >> 11) * 2.0**-53
    return float(np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(6.283185307179586 * u2))


def lower_factor(C: np.ndarray) -> np.ndarray:
    """Lower-triangular L with L L' = C for PSD C, right-looking; a non-positive pivot zeroes its column
    (the rule of the device kernels, ek_math.h chol_packed).  Same distribution as any other square root."""
    A = np.array(C, float)
    n = A.shape[0]
    L = np.zeros_like(A)
    for k in range(n):
        piv = A[k, k]
        if piv > 0.0:
            L[k, k] = np.sqrt(piv)
            L[k + 1:, k] = A[k + 1:, k] / L[k, k]
            A[k + 1:, k + 1:] -= np.outer(L[k + 1:, k], L[k + 1:, k])
    return L


def sample_states_on(ts, xs, diffusions, difftimes, consts, n: int = 1, seed: int = 0x5A3B1E, traj: int = 0,
                     sqrt: str = "reference", noise_scale: float = 1.0, normal=None) -> np.ndarray:
    """solution_sampling.jl:24-62 for states `xs` at times `ts`: draw x_N ~ N(mu_N, S_N), then backwards
    x_i ~ smooth(xs[i], delta(x_{i+1})) in preconditioned coordinates.  Returns [len(ts), D, n].
    sqrt = "reference": the square root the reference multiplies the noise with (R' of its QR, :10);
    sqrt = "cholesky":  the lower-triangular factor of the same covariance (what the device uses)."""
    A, Q_L, precond, d, q = consts
    D = d * (q + 1)
    ns = len(xs)
    assert len(diffusions) + 1 == len(difftimes)  # :25
    normal = normal or (lambda j, slot, k: sample_normal(seed, traj, j, slot, k, n, ns, D))
    path = np.zeros((ns, D, n))
    difftimes = np.asarray(difftimes, float)

    def draw(g: SRGaussian, j: int, slot: int) -> np.ndarray:
        xi = np.array([normal(j, slot, k) for k in range(D)])
        S = g.L if sqrt == "reference" else lower_factor(g.L @ g.L.T)
        return g.mu + noise_scale * (S @ xi)  # _rand, :6-12

    for j in range(n):
        path[ns - 1, :, j] = draw(xs[-1], j, ns - 1)
    for i in range(ns - 2, -1, -1):  # Julia i = length(xs)-1 .. 1
        dt = ts[i + 1] - ts[i]
        i_diffusion = int(np.sum(difftimes <= ts[i]))  # 1-based (:41)
        diffusion = diffusions[i_diffusion - 1]
        Qh = apply_diffusion(Q_L, diffusion)
        P = precond(dt)
        PI = 1.0 / P
        for j in range(n):
            nxt = SRGaussian(P * path[i + 1, :, j], np.zeros((D, D)))
            prev_p, _ = smooth(linmap(P, xs[i]), nxt, A, Qh)
            if sqrt == "reference":
                path[i, :, j] = PI * draw(prev_p, j, i)
            else:  # un-precondition first: chol(PI C PI) = PI chol(C), same sample up to rounding
                path[i, :, j] = draw(linmap(PI, prev_p), j, i)
    return path


def sample_states(sol: Solution, consts, n: int = 1, **kw) -> np.ndarray:
    """solution_sampling.jl:15-18: on the saved grid, from the filter states."""
    return sample_states_on(sol.t, sol.x_filt, sol.diffusions, sol.t, consts, n, **kw)


def dense_sample_states(sol: Solution, consts, n: int = 1, times=None, **kw):
    """solution_sampling.jl:63-69: sampling on a dense grid (1 000 points by default) of FILTER-interpolated states."""
    times = np.linspace(sol.t[0], sol.t[-1], 1000) if times is None else np.asarray(times, float)
    states = [dense_output(sol, consts, float(t), smoothed=False) for t in times]
    return sample_states_on(times, states, sol.diffusions, sol.t, consts, n, **kw), times


def sample(sol: Solution, consts, n: int = 1, **kw) -> np.ndarray:
    """solution_sampling.jl:19-23: the zeroth-derivative part [n_save, d, n]."""
    d = consts[3]
    return sample_states(sol, consts, n, **kw)[:, :d, :]


def dense_sample(sol: Solution, consts, n: int = 1, times=None, **kw):
    """solution_sampling.jl:70-75: the zeroth-derivative part of dense_sample_states, and the times."""
    d = consts[3]
    states, times = dense_sample_states(sol, consts, n, times=times, **kw)
    return states[:, :d, :], times


# --------------------------------------------------------------------------------------
# Ensemble inputs: bit-identical in Python / C / HIP
# --------------------------------------------------------------------------------------

_M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def ensemble_u0(base_u0: np.ndarray, n_traj: int, scale: float, seed: int = 0x0DEF17E5,
                first: int = 0, n_perturbed: Optional[int] = None) -> np.ndarray:
    """u0_i[k] = base[k] + scale*(2U-1), U = (splitmix64(seed + d_p*i + k) >> 11) * 2^-53.
    Returns [n_traj, d].  `n_perturbed` = number of leading components perturbed."""
    d = len(base_u0)
    dp = d if n_perturbed is None else n_perturbed
    out = np.tile(np.asarray(base_u0, float), (n_traj, 1))
    for ii in range(n_traj):
        i = first + ii
        for k in range(dp):
            U = (splitmix64((seed + dp * i + k) & _M64) >> 11) * 2.0**-53
            out[ii, k] = base_u0[k] + scale * (2.0 * U - 1.0)
    return out
