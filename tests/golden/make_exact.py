"""High-precision evaluation of the reference ALGORITHM (not of any float64 implementation of it): what do the EK1
filter and the RTS smoother of src/perform_step.jl / src/filtering.jl / src/smoothing.jl give in (near-)exact arithmetic?

The tolerances of the parity tests rest on this.  Float64 implementations of the reference arithmetic agree in the
solution components u = E0 mu to 1e-14, but NOT in the higher-derivative blocks of the state and in the covariances:
those are ill-conditioned in the reference's own formulation (the float64 oracle itself is 1e-8 / 3e-4 away from the
exact result there).  So "the device agrees with the oracle to x" is the wrong question for those blocks; the right one is
"is the device as close to the exact result as the oracle is".  This script computes the exact results:

  exact_lorenz_mp.npz    Lorenz-63, EK1(order=3), dt = 2^-9, 1 024 steps (BASELINE configs 2/3, trajectory 0 of the
                         ensemble), filter AND smoother, mpmath with 50 significant digits
  exact_pleiades_ld.npz  Pleiades, EK1(order=5), dt = 2^-10, 24 steps (BASELINE config 4 at test size, trajectory 0),
                         filter, numpy longdouble (x87 extended, 64-bit mantissa = 2 048 x finer than float64: D = 168
                         is out of reach for mpmath in a build-container minute, and 11 extra bits are enough to rank
                         two float64 results)
  exact_pleiades_{ek1q2,ek0q3,ek1q5,ek1q5_dt6}_smooth_ld.npz
                         Pleiades, 12 steps, trajectories 0 and 4, filter AND RTS smoother (D = 84, 112, 168), longdouble: what
                         the D = 168 kernels' parity tests are measured against (round 3; until then their tolerance was
                         calibrated on the float64 oracle's own spread, and the order-5 covariances were not compared at all)

and, next to them, the float64 oracle's distance from them per derivative block and for the covariance.  Any
mathematically equivalent formulation gives the same exact result, so the recursion is written in the plain covariance
form (Sigma^- = A Sigma A' + sigma^2 Q, K = Sigma^- H' S^-1, Sigma = (I - K H) Sigma^- (I - K H)'; G = Sigma A' (Sigma^-)^-1)
in preconditioned coordinates.  (The Joseph form of the update matters even at 50 digits: Sigma^- - K S K' amplifies
rounding errors by ~6x per step on this problem and is useless after 60 steps.)

Run (build container, ~3 min):  python tests/golden/make_exact.py
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))
import odefilter_oracle as orc  # noqa: E402
from _parity import block_err, cov_err  # noqa: E402


# ---------------------------------------------------------------------------------------------- mpmath, Lorenz-63
def lorenz_mp(u0, p, q, dt, nsteps, digits=50):
    import mpmath as mp

    mp.mp.dps = digits
    d, NB = 3, q + 1
    D = d * NB
    M = mp.matrix
    h = mp.mpf(dt)
    sig, rho, beta = [mp.mpf(float(x)) for x in p]  # the float64 parameter values, exactly

    def f(u):
        return [sig * (u[1] - u[0]), u[0] * (rho - u[2]) - u[1], u[0] * u[1] - beta * u[2]]

    def jac(u):
        return M([[-sig, sig, 0], [rho - u[2], -1, -u[0]], [u[1], u[0], -beta]])

    # Taylor-mode initial state (src/state_initialization.jl:2-53): series coefficients by the Lie recursion
    c = [[mp.mpf(float(x))] + [mp.mpf(0)] * q for x in u0]  # c[a][k]
    for k in range(q):
        def conv(x, y, n):
            return sum(x[j] * y[n - j] for j in range(n + 1))
        fx = sig * (c[1][k] - c[0][k])
        fy = rho * c[0][k] - conv(c[0], c[2], k) - c[1][k]
        fz = conv(c[0], c[1], k) - beta * c[2][k]
        for a, v in enumerate((fx, fy, fz)):
            c[a][k + 1] = v / (k + 1)
    m = M(D, 1)
    for k in range(NB):
        for a in range(d):
            m[k * d + a] = c[a][k] * mp.factorial(k)
    # prior (src/priors.jl:7-59) and preconditioner (src/preconditioning.jl:1-17)
    A = mp.eye(D)
    for i in range(1, q + 1):
        for j in range(d * (q + 1 - i)):
            A[j, j + d * i] = mp.mpf(1) / mp.factorial(i)
    Q = M(D, D)
    for col in range(NB):
        for row in range(NB):
            v = mp.mpf(1) / ((2 * q + 1 - row - col) * mp.factorial(q - row) * mp.factorial(q - col))
            for i in range(d):
                Q[row * d + i, col * d + i] = v
    Pd = [h ** (mp.mpf(j) - q - mp.mpf(1) / 2) for j in range(NB) for _ in range(d)]
    E0 = M(d, D)
    E1 = M(d, D)
    for a in range(d):
        E0[a, a] = 1
        E1[a, d + a] = 1
    Pm = mp.diag(Pd)
    PIm = mp.diag([1 / x for x in Pd])
    S_ = M(D, D)  # Sigma_0 = 0
    means, covs, diffs = [m.copy()], [S_.copy()], []
    for _ in range(nsteps):
        mt, X = Pm * m, Pm * S_ * Pm
        mp_ = A * mt
        up = E0 * (PIm * mp_)
        z = E1 * (PIm * mp_) - M(f([up[0], up[1], up[2]]))
        H = (E1 - jac([up[0], up[1], up[2]]) * E0) * PIm
        W = H * Q * H.T
        s2 = (z.T * mp.lu_solve(W, z))[0] / d
        Xp = A * X * A.T + s2 * Q
        C = Xp * H.T
        Sm = H * C
        K = C * mp.inverse(Sm)  # K = C S^-1 (50 digits: the explicit inverse is harmless)
        mf = mp_ - K * z
        IKH = mp.eye(D) - K * H
        Xf = IKH * Xp * IKH.T  # Joseph form: the plain Sigma^- - K S K' amplifies rounding errors ~6x per step (50 digits last 60 steps)
        m, S_ = PIm * mf, PIm * Xf * PIm
        means.append(m.copy()); covs.append(S_.copy()); diffs.append(s2)
    # RTS pass (src/smoothing.jl:4-63)
    sm, sc = [None] * (nsteps + 1), [None] * (nsteps + 1)
    sm[nsteps], sc[nsteps] = means[nsteps], covs[nsteps]
    sm[0], sc[0] = means[0], covs[0]
    for i in range(nsteps - 1, 0, -1):
        mt, X = Pm * means[i], Pm * covs[i] * Pm
        mpred = A * mt
        B = A * X * A.T + diffs[i] * Q
        G = X * A.T * mp.inverse(B)
        ms = mt + G * (Pm * sm[i + 1] - mpred)
        Ss = X + G * (Pm * sc[i + 1] * Pm - B) * G.T
        sm[i], sc[i] = PIm * ms, PIm * Ss * PIm
    tof = lambda v: np.array([float(x) for x in v])  # noqa: E731
    tom = lambda a: np.array([[float(a[i, j]) for j in range(D)] for i in range(D)])  # noqa: E731
    return (np.array([tof(x) for x in means]), np.array([tom(x) for x in covs]), np.array([tof(x) for x in sm]),
            np.array([tom(x) for x in sc]), np.array([float(x) for x in diffs]))


# ------------------------------------------------------------------------------ longdouble, any vector field (filter)
def chol_ld(S):
    n = S.shape[0]
    L = np.zeros_like(S)
    for j in range(n):
        s = S[j, j] - L[j, :j] @ L[j, :j]
        L[j, j] = np.sqrt(s)
        L[j + 1:, j] = (S[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def solve_spd_ld(S, B):
    """S^-1 B for SPD S, longdouble."""
    L = chol_ld(S)
    n = S.shape[0]
    Y = np.zeros_like(B)
    for i in range(n):
        Y[i] = (B[i] - L[i, :i] @ Y[:i]) / L[i, i]
    X = np.zeros_like(B)
    for i in range(n - 1, -1, -1):
        X[i] = (Y[i] - L[i + 1:, i] @ X[i + 1:]) / L[i, i]
    return X


class JetLD:
    """Truncated Taylor arithmetic on longdouble coefficients (the oracle's Jet is float64 by construction): just what the
    Pleiades field needs -- +, -, *, real powers."""

    def __init__(self, c):
        self.c = np.asarray(c, dtype=np.longdouble)

    @staticmethod
    def lift(x, n):
        if isinstance(x, JetLD):
            return x
        c = np.zeros(n, dtype=np.longdouble)
        c[0] = x
        return JetLD(c)

    def __add__(self, o):
        return JetLD(self.c + JetLD.lift(o, len(self.c)).c)

    __radd__ = __add__

    def __sub__(self, o):
        return JetLD(self.c - JetLD.lift(o, len(self.c)).c)

    def __rsub__(self, o):
        return JetLD(JetLD.lift(o, len(self.c)).c - self.c)

    def __mul__(self, o):
        if not isinstance(o, JetLD):
            return JetLD(self.c * np.longdouble(o))
        n = len(self.c)
        return JetLD([np.dot(self.c[: k + 1], o.c[k::-1]) for k in range(n)])

    __rmul__ = __mul__

    def __pow__(self, a):
        n = len(self.c)
        out = np.zeros(n, dtype=np.longdouble)
        out[0] = self.c[0] ** np.longdouble(a)
        for k in range(1, n):
            s = np.longdouble(0)
            for j in range(1, k + 1):
                s += (np.longdouble(a) * j - (k - j)) * self.c[j] * out[k - j]
            out[k] = s / (k * self.c[0])
        return JetLD(out)


def pleiades_jac_ld(u):
    ld = np.longdouble
    x, y = u[0:7], u[7:14]
    J = np.zeros((28, 28), dtype=ld)
    J[0:7, 14:21] = np.eye(7)
    J[7:14, 21:28] = np.eye(7)
    for i in range(7):
        for j in range(7):
            if j == i:
                continue
            mj = ld(j + 1)
            dx, dy = x[j] - x[i], y[j] - y[i]
            r2 = dx * dx + dy * dy
            r3, r5 = r2 ** ld(-1.5), r2 ** ld(-2.5)
            axx = mj * (r3 - 3 * dx * dx * r5)
            axy = mj * (-3 * dx * dy * r5)
            ayy = mj * (r3 - 3 * dy * dy * r5)
            J[14 + i, j] += axx; J[14 + i, i] -= axx; J[14 + i, 7 + j] += axy; J[14 + i, 7 + i] -= axy
            J[21 + i, j] += axy; J[21 + i, i] -= axy; J[21 + i, 7 + j] += ayy; J[21 + i, 7 + i] -= ayy
    return J


def filter_ld(vf, u0, q, dt, nsteps, kind="EK1", with_smoother=False):
    """Pleiades only (its f is written for any scalar type, its Jacobian is restated above in longdouble).  EK0 / EK1 filter and,
    if asked, the RTS pass (src/smoothing.jl:4-63) over the same records, all in longdouble.  Returns float64 copies:
    (means, covs) or (means, covs, smoothed means, smoothed covs, diffusions)."""
    ld = np.longdouble
    d, NB = vf.d, q + 1
    D = d * NB
    coef = np.zeros((d, NB), dtype=ld)
    coef[:, 0] = np.asarray(u0, dtype=ld)
    for k in range(q):  # Taylor-mode initial state (src/state_initialization.jl:15-42) in longdouble
        fu = vf.f([JetLD(coef[i].copy()) for i in range(d)], None, 0.0)
        for i in range(d):
            coef[i, k + 1] = JetLD.lift(fu[i], NB).c[k] / (k + 1)
    m = np.concatenate([coef[:, k] * ld(math.factorial(k)) for k in range(NB)])
    A = np.eye(D, dtype=ld)
    for i in range(1, q + 1):
        for j in range(d * (q + 1 - i)):
            A[j, j + d * i] = ld(1) / ld(math.factorial(i))
    Q = np.zeros((D, D), dtype=ld)
    for col in range(NB):
        for row in range(NB):
            v = ld(1) / (ld(2 * q + 1 - row - col) * ld(math.factorial(q - row)) * ld(math.factorial(q - col)))
            for i in range(d):
                Q[row * d + i, col * d + i] = v
    h = ld(dt)
    P = np.repeat(np.array([h ** (ld(j) - ld(q) - ld(0.5)) for j in range(NB)], dtype=ld), d)
    PI = ld(1) / P
    S_ = np.zeros((D, D), dtype=ld)
    means, covs, diffs = [m.copy()], [S_.copy()], []
    for n in range(nsteps):
        mt, X = P * m, S_ * np.outer(P, P)
        mp_ = A @ mt
        up = (PI * mp_)[:d]
        du = np.asarray(vf.f(list(up), None, 0.0), dtype=ld)
        z = (PI * mp_)[d:2 * d] - du
        H = np.zeros((d, D), dtype=ld)
        if kind == "EK1":
            H[:, :d] = -pleiades_jac_ld(up) * PI[:d][None, :]
        H[:, d:2 * d] = np.diag(PI[d:2 * d])
        W = H @ Q @ H.T
        s2 = (z @ solve_spd_ld(W, z[:, None])[:, 0]) / d
        Xp = A @ X @ A.T + s2 * Q
        C = Xp @ H.T
        Sm = H @ C
        K = solve_spd_ld(Sm, C.T).T
        mf = mp_ - K @ z
        IKH = np.eye(D, dtype=ld) - K @ H
        Xf = IKH @ Xp @ IKH.T  # Joseph form (see lorenz_mp)
        Xf = (Xf + Xf.T) / 2
        m, S_ = PI * mf, Xf * np.outer(PI, PI)
        means.append(m.copy()); covs.append(S_.copy()); diffs.append(s2)
    f64 = lambda a: np.array(a).astype(np.float64)  # noqa: E731
    if not with_smoother:
        return f64(means), f64(covs)
    # RTS pass: x_i^s from x_i (filter) and x_{i+1}^s, G = X A' (A X A' + sigma_i^2 Q)^-1 in preconditioned coordinates;
    # diffusions[i] belongs to the step t_i -> t_{i+1} (src/integrator_utils.jl:44)
    sm, sc = [None] * (nsteps + 1), [None] * (nsteps + 1)
    sm[nsteps], sc[nsteps] = means[nsteps], covs[nsteps]
    sm[0], sc[0] = means[0], covs[0]
    PP = np.outer(P, P)
    for i in range(nsteps - 1, 0, -1):
        mt, X = P * means[i], covs[i] * PP
        B = A @ X @ A.T + diffs[i] * Q
        B = (B + B.T) / 2
        G = solve_spd_ld(B, A @ X).T  # X A' B^-1 (X, B symmetric)
        ms_ = mt + G @ (P * sm[i + 1] - A @ mt)
        Ss = X + G @ (sc[i + 1] * PP - B) @ G.T
        Ss = (Ss + Ss.T) / 2
        sm[i], sc[i] = PI * ms_, Ss * np.outer(PI, PI)
    return f64(means), f64(covs), f64(sm), f64(sc), f64(diffs)


def pleiades_fixture(q, kind, nsteps, dt, trajs=(0, 4)):
    """Trajectories `trajs` of the Pleiades ensemble (positions perturbed by 1e-3): filter and smoother in
    extended precision, and the float64 oracle's distance from them -- per derivative block for the means (all records) and
    for the covariance at one record each: the LAST filter record and smoothed record 1 (the one the backward pass reaches
    last), stored as packed lower triangles."""
    vp = orc.vector_field("pleiades")
    u0s = orc.ensemble_u0(vp.u0, max(trajs) + 1, 1e-3, n_perturbed=14)
    D = 28 * (q + 1)
    il = np.tril_indices(D)
    out = dict(trajs=np.array(trajs), u0s=u0s[list(trajs)], dt=dt, nsteps=nsteps, order=q, ek1=int(kind == "EK1"),
               cov_record_filt=nsteps, cov_record_smooth=1)
    acc = {k: [] for k in ("mean_filt", "mean_smooth", "diffusions", "cov_filt_tril", "cov_smooth_tril", "oracle_block_err_filt",
                           "oracle_block_err_smooth", "oracle_cov_err_filt", "oracle_cov_err_smooth", "oracle_diffusion_err")}
    for i in trajs:
        mf, cf, ms, cs, df = filter_ld(vp, u0s[i], q, dt, nsteps, kind=kind, with_smoother=True)
        sol = orc.solve(vp, orc.Alg(kind, q, "dynamic", True), u0=u0s[i], tspan=(0.0, nsteps * dt), dt=dt)
        acc["mean_filt"].append(mf); acc["mean_smooth"].append(ms); acc["diffusions"].append(df)
        acc["cov_filt_tril"].append(cf[nsteps][il]); acc["cov_smooth_tril"].append(cs[1][il])
        acc["oracle_block_err_filt"].append(block_err(sol.means(smoothed=False), mf, 28))
        acc["oracle_block_err_smooth"].append(block_err(sol.means(smoothed=True), ms, 28))
        acc["oracle_cov_err_filt"].append(cov_err(sol.covs(smoothed=False)[nsteps:nsteps + 1], cf[nsteps:nsteps + 1]))
        acc["oracle_cov_err_smooth"].append(cov_err(sol.covs(smoothed=True)[1:2], cs[1:2]))
        acc["oracle_diffusion_err"].append(float(np.max(np.abs(sol.diffusions - df) / np.abs(df))))
    out.update({k: np.array(v) for k, v in acc.items()})
    return out


if __name__ == "__main__":
    vf = orc.vector_field("lorenz63")
    u0 = orc.ensemble_u0(vf.u0, 1, 1e-2)[0]
    dt, ns = 2.0**-9, 1024
    mf, cf, ms, cs, df = lorenz_mp(u0, vf.p, 3, dt, ns)
    sol = orc.solve(vf, orc.EK1(order=3, smooth=True), u0=u0, tspan=(0.0, ns * dt), dt=dt)
    out = dict(u0=u0, dt=dt, nsteps=ns, digits=50, mean_filt=mf, cov_filt=cf, mean_smooth=ms, cov_smooth=cs, diffusions=df,
               oracle_block_err_filt=block_err(sol.means(smoothed=False), mf, 3), oracle_cov_err_filt=cov_err(sol.covs(smoothed=False), cf),
               oracle_block_err_smooth=block_err(sol.means(smoothed=True), ms, 3), oracle_cov_err_smooth=cov_err(sol.covs(smoothed=True), cs))
    np.savez_compressed(os.path.join(HERE, "exact_lorenz_mp.npz"), **out)
    print("exact_lorenz_mp: oracle vs exact, filter blocks", out["oracle_block_err_filt"], "cov", out["oracle_cov_err_filt"],
          "| smoother blocks", out["oracle_block_err_smooth"], "cov", out["oracle_cov_err_smooth"])

    vp = orc.vector_field("pleiades")
    u0p = orc.ensemble_u0(vp.u0, 1, 1e-3, n_perturbed=14)[0]
    dtp, nsp = 2.0**-10, 24
    mfp, cfp = filter_ld(vp, u0p, 5, dtp, nsp)
    solp = orc.solve(vp, orc.EK1(order=5, smooth=False), u0=u0p, tspan=(0.0, nsp * dtp), dt=dtp)
    outp = dict(u0=u0p, dt=dtp, nsteps=nsp, mean_filt=mfp, var_filt=np.array([np.diag(c) for c in cfp]), cov_final=cfp[-1],
                oracle_block_err_filt=block_err(solp.means(smoothed=False), mfp, 28),
                oracle_cov_err_final=cov_err(solp.covs(smoothed=False)[-1:], cfp[-1:]))
    np.savez_compressed(os.path.join(HERE, "exact_pleiades_ld.npz"), **outp)
    print("exact_pleiades_ld: oracle vs extended precision, filter blocks", outp["oracle_block_err_filt"], "final cov", outp["oracle_cov_err_final"])

    # filter + smoother at the sizes of the GPU parity tests (tests/test_gpu_parity.py test_pleiades_ensemble_parity: 12 steps of
    # config 4's dt = 2^-10), and order 5 once more with dt = 2^-6: at 2^-10 the residuals of the first steps of an order-5 solve are
    # h^5-small, i.e. rounding noise in float64 (the oracle's own diffusions are 40 % off there) -- the larger step is where
    # an order-5 covariance can be checked sharply
    for q_, kind_, dt_, tag in ((2, "EK1", 2.0**-10, ""), (3, "EK0", 2.0**-10, ""), (5, "EK1", 2.0**-10, ""), (5, "EK1", 2.0**-6, "_dt6")):
        fxp = pleiades_fixture(q_, kind_, 12, dt_)
        name = f"exact_pleiades_{kind_.lower()}q{q_}{tag}_smooth_ld"
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fxp)
        print(name, ": oracle vs extended precision, filter blocks", fxp["oracle_block_err_filt"].max(axis=0), "cov", fxp["oracle_cov_err_filt"],
              "| smoother blocks", fxp["oracle_block_err_smooth"].max(axis=0), "cov", fxp["oracle_cov_err_smooth"], "| diffusions", fxp["oracle_diffusion_err"])
