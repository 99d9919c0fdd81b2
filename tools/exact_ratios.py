#!/usr/bin/env python3
"""How far the D = 84 / 112 / 168 kernels are from the extended-precision evaluation of the reference algorithm, in units of the
float64 oracle's own distance (tests/golden/exact_pleiades_*_smooth_ld.npz, generator tests/golden/make_exact.py): one JSON line
per fixture and kernel combination -- the numbers behind the factor-16 bar of tests/_parity.py check_against_exact_fixture."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import odefilters_jl_amd as pkg
import odefilter_oracle as orc
import _parity as P

GOLD = os.path.join(ROOT, "tests", "golden")
vf = orc.vector_field("pleiades")
for q, kind, tag in ((2, "EK1", ""), (3, "EK0", ""), (5, "EK1", ""), (5, "EK1", "_dt6")):
    fx = np.load(os.path.join(GOLD, f"exact_pleiades_{kind.lower()}q{q}{tag}_smooth_ld.npz"))
    ns, dt = int(fx["nsteps"]), float(fx["dt"])
    D = 28 * (q + 1)
    il = np.tril_indices(D)
    for kernels in ("mfma+split", "mfma+persistent", "tiles+split"):
        filt, smoother = kernels.split("+")
        os.environ["ODEF_PLEIADES_FILTER"] = "tiles" if filt == "tiles" else ""
        os.environ["ODEF_SMOOTH_SPLIT"] = "1" if smoother == "split" else "0"
        ens = pkg.EnsembleProblem(pkg.ODEProblem("pleiades", vf.u0, (0.0, ns * dt), ()), perturb_scale=1e-3, n_perturbed=14)
        alg = (pkg.EK1 if kind == "EK1" else pkg.EK0)(order=q, smooth=True)
        sol = pkg.solve(ens, alg, pkg.EnsembleHIP(), trajectories=5, dt=dt, adaptive=False)
        out = {"fixture": f"{kind}({q}){tag}", "dt": dt, "kernels": kernels}
        for name, mean, cov, rec in (("filt", sol.x_filt_mean(), sol.x_filt_cov(), ns), ("smooth", sol.x_smooth_mean(), sol.x_smooth_cov(), 1)):
            yard_b = np.maximum(fx["oracle_block_err_" + name].max(axis=0), 1e-16)
            yard_c = float(fx["oracle_cov_err_" + name].max())
            rb, rc = [], []
            for k, i in enumerate(fx["trajs"]):
                rb.append(P.block_err(mean[i], fx["mean_" + name][k], 28) / yard_b)
                ec = np.zeros((D, D)); ec[il] = fx[f"cov_{name}_tril"][k]; ec = ec + np.tril(ec, -1).T
                rc.append(P.cov_err(cov[i][rec][None], ec[None]) / yard_c)
            out[name] = {"block_ratio": [round(float(x), 2) for x in np.max(rb, axis=0)], "cov_ratio": round(float(max(rc)), 2),
                         "oracle_block_err": [float(f"{x:.2e}") for x in yard_b], "oracle_cov_err": float(f"{yard_c:.2e}")}
        print(json.dumps(out), flush=True)
