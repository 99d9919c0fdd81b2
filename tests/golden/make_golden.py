"""Generates tests/golden/*.npz from the numpy oracle (oracle/odefilter_oracle.py).

The reference (Julia) cannot run in the build image, so these vectors are outputs of the
oracle, which is itself pinned to the reference's own tests (tests/test_oracle_kats.py).
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import odefilter_oracle as orc  # noqa: E402


def run_case(name, rhs, alg, *, n_traj, scale, dt=None, tspan=None, adaptive=False, abstol=1e-6, reltol=1e-3):
    vf = orc.vector_field(rhs)
    tspan = tspan or vf.tspan
    u0s = orc.ensemble_u0(vf.u0, n_traj, scale) if n_traj > 1 else vf.u0[None, :].copy()
    out = dict(rhs=rhs, kind=alg.kind, order=alg.order, diffusionmodel=alg.diffusionmodel, smooth=alg.smooth,
               u0s=u0s, p=vf.p, tspan=np.array(tspan), dt=np.array(dt if dt is not None else np.nan),
               adaptive=adaptive, abstol=abstol, reltol=reltol, scale=scale)
    ts, mf, cf, ms, cs, df, ll, na, nr = [], [], [], [], [], [], [], [], []
    for i in range(n_traj):
        sol = orc.solve(vf, alg, u0=u0s[i], tspan=tspan, dt=dt, adaptive=adaptive, abstol=abstol, reltol=reltol)
        ts.append(np.array(sol.t)); mf.append(sol.means(smoothed=False)); cf.append(sol.covs(smoothed=False))
        if alg.smooth:
            ms.append(sol.means(smoothed=True)); cs.append(sol.covs(smoothed=True))
        df.append(np.array(sol.diffusions)); ll.append(sol.log_likelihood); na.append(sol.naccept); nr.append(sol.nreject)
    if adaptive:  # ragged: pad with NaN
        L = max(len(t) for t in ts)

        def pad(a):
            return np.array([np.concatenate([x, np.full((L - len(x),) + x.shape[1:], np.nan)]) for x in a])

        ts, mf, cf, df = pad(ts), pad(mf), pad(cf), pad([np.concatenate([[0.0], d]) for d in df])[:, 1:]
        if alg.smooth:
            ms, cs = pad(ms), pad(cs)
    out.update(t=np.array(ts), mean_filt=np.array(mf), cov_filt=np.array(cf), diffusions=np.array(df),
               loglik=np.array(ll), naccept=np.array(na), nreject=np.array(nr))
    if alg.smooth:
        out.update(mean_smooth=np.array(ms), cov_smooth=np.array(cs))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", out["mean_filt"].shape)


def run_pleiades(name, n_traj=2, nsteps=24, q=5):
    """BASELINE config 4 at test size: final filter state + u time series only (a full record is 226 KB per step)."""
    vf = orc.vector_field("pleiades")
    dt = 2.0**-10
    u0s = orc.ensemble_u0(vf.u0, n_traj, 1e-3, n_perturbed=14)
    mf, cf, us, df, ll = [], [], [], [], []
    for i in range(n_traj):
        sol = orc.solve(vf, orc.EK1(order=q, smooth=False), u0=u0s[i], tspan=(0.0, nsteps * dt), dt=dt)
        mf.append(sol.x_filt[-1].mu); cf.append(sol.x_filt[-1].cov()); us.append(sol.u); df.append(sol.diffusions); ll.append(sol.log_likelihood)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), u0s=u0s, nsteps=nsteps, dt=dt, order=q, mean_final=np.array(mf),
                        cov_final=np.array(cf).astype(np.float64), u=np.array(us), diffusions=np.array(df), loglik=np.array(ll))
    print(name, "ok")


if __name__ == "__main__":
    run_pleiades("pleiades_ek1_q5_cfg4")
    # config 1 (BASELINE.json): FHN, EK0(order=1), single trajectory, dt = 7e-2, smooth
    run_case("fhn_ek0_q1_cfg1", "fhn", orc.EK0(order=1), n_traj=1, scale=0.0, dt=7e-2)
    # config 2/3 shape at test size: Lorenz-63 EK1(order=3), perturbed ensemble, dt = 2^-9
    run_case("lorenz_ek1_q3", "lorenz63", orc.EK1(order=3), n_traj=4, scale=1e-2, dt=2.0**-9, tspan=(0.0, 0.25))
    run_case("lv_ek1_q2_fixeddiff", "lotka_volterra", orc.EK1(order=2, diffusionmodel="fixed"), n_traj=2, scale=1e-2,
             dt=5e-3, tspan=(0.0, 0.5))
    run_case("lv_ek0_q4", "lotka_volterra", orc.EK0(order=4), n_traj=2, scale=1e-2, dt=1e-2, tspan=(0.0, 0.5))
    run_case("vdp_ek1_q5", "vanderpol", orc.EK1(order=5), n_traj=1, scale=0.0, dt=2e-2, tspan=(0.0, 1.0))
    # config 5 shape at test size: adaptive PI + smoothing
    run_case("lorenz_ek1_q3_adaptive", "lorenz63", orc.EK1(order=3), n_traj=3, scale=1e-2, dt=2.0**-9,
             tspan=(0.0, 0.5), adaptive=True)
