"""Full-size fixtures: a handful of trajectories of the BASELINE.json ensembles (configs 2-5) run through the numpy
oracle (oracle/odefilter_oracle.py) at the FULL step count, so that the GPU tests and bench.py can compare device
results of full-size runs with committed numbers (the oracle itself is far too slow for the whole ensembles).
The ensembles are the splitmix64 ones of SURVEY.md 8(d): trajectory i of an N-trajectory ensemble does not depend on N.

Run (build container, ~2 min):  python tests/golden/make_fullsize.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import odefilter_oracle as orc  # noqa: E402


def u0_of(vf, idx, scale, n_perturbed=None):
    n = max(idx) + 1
    kw = {} if n_perturbed is None else dict(n_perturbed=n_perturbed)
    return orc.ensemble_u0(vf.u0, n, scale, **kw)[list(idx)]


def lorenz_fixed():
    """configs 2 and 3: Lorenz-63 EK1(3), dt = 2^-9, 1 024 steps; filter + smoother (config 2), final state (config 3)."""
    vf = orc.vector_field("lorenz63")
    idx = [0, 1, 63, 64, 4095, 4097, 8191]  # all < 8 192: on rank 0 for every shard count up to 8; < 4 096 for config 2
    u0s = u0_of(vf, idx, 1e-2)
    dt, ns = 2.0**-9, 1024
    steps = [0, 1, 256, 512, 777, 1023, 1024]
    mf, cf, ms, cs, df, ll = [], [], [], [], [], []
    for u0 in u0s:
        sol = orc.solve(vf, orc.EK1(order=3, smooth=True), u0=u0, tspan=(0.0, ns * dt), dt=dt)
        mf.append(sol.means(smoothed=False)[steps]); cf.append(sol.covs(smoothed=False)[steps])
        ms.append(sol.means(smoothed=True)[steps]); cs.append(sol.covs(smoothed=True)[steps])
        df.append(np.array(sol.diffusions)); ll.append(sol.log_likelihood)
    np.savez_compressed(os.path.join(HERE, "full_lorenz_fixed.npz"), idx=np.array(idx), u0s=u0s, steps=np.array(steps),
                        mean_filt=np.array(mf), cov_filt=np.array(cf), mean_smooth=np.array(ms), cov_smooth=np.array(cs),
                        diffusions=np.array(df), loglik=np.array(ll), dt=dt, nsteps=ns)
    print("full_lorenz_fixed ok")


def lorenz_adaptive():
    """config 5: Lorenz-63 EK1(3), adaptive PI (abstol 1e-6, reltol 1e-3, dt0 = 2^-9), t in [0, 2], RTS smoothing."""
    vf = orc.vector_field("lorenz63")
    idx = [0, 1, 4097, 16383]
    u0s = u0_of(vf, idx, 1e-2)
    out = dict(idx=np.array(idx), u0s=u0s)
    for k, u0 in enumerate(u0s):
        sol = orc.solve(vf, orc.EK1(order=3, smooth=True), u0=u0, tspan=(0.0, 2.0), dt=2.0**-9, adaptive=True)
        out[f"t{k}"] = np.array(sol.t)
        out[f"mean_filt{k}"] = sol.means(smoothed=False)
        out[f"mean_smooth{k}"] = sol.means(smoothed=True)
        out[f"var_smooth{k}"] = np.array([np.diag(c) for c in sol.covs(smoothed=True)])
        out[f"counts{k}"] = np.array([sol.naccept, sol.nreject])
    np.savez_compressed(os.path.join(HERE, "full_lorenz_adaptive.npz"), **out)
    print("full_lorenz_adaptive ok")


def pleiades():
    """config 4: Pleiades EK1(5), dt = 2^-10, 256 steps, final state."""
    vf = orc.vector_field("pleiades")
    idx = [0, 8191]
    u0s = u0_of(vf, idx, 1e-3, n_perturbed=14)
    dt, ns = 2.0**-10, 256
    mf, vf_, us = [], [], []
    for u0 in u0s:
        sol = orc.solve(vf, orc.EK1(order=5, smooth=False), u0=u0, tspan=(0.0, ns * dt), dt=dt)
        mf.append(sol.x_filt[-1].mu); vf_.append(np.diag(sol.x_filt[-1].cov())); us.append(sol.u[-1])
    np.savez_compressed(os.path.join(HERE, "full_pleiades.npz"), idx=np.array(idx), u0s=u0s, mean_final=np.array(mf),
                        var_final=np.array(vf_), u_final=np.array(us), dt=dt, nsteps=ns)
    print("full_pleiades ok")


if __name__ == "__main__":
    lorenz_fixed()
    lorenz_adaptive()
    pleiades()
