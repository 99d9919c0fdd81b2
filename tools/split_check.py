#!/usr/bin/env python3
"""A/B of the D = 28 (q+1) smoother: persistent kernel against the split pass (ODEF_SMOOTH_SPLIT=1), fixed grid and adaptive."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odefilters_jl_amd as pkg
from oracle import odefilter_oracle as orc
vf = orc.vector_field("pleiades")
q = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 70
adaptive = len(sys.argv) > 3 and sys.argv[3] == "adaptive"
out = {}
for name, env in (("persistent", "0"), ("split", "1")):  # (split is the default)
    os.environ["ODEF_SMOOTH_SPLIT"] = env
    ctx = pkg.Context("pleiades", q, 1, N, save_everystep=True)
    ctx.set_problem_perturbed(vf.u0, [], 0.0, 3e-2 if adaptive else 1e-3, n_perturbed=14)
    if adaptive:
        ctx.solve_adaptive(0.06, 1e-7, 1e-5, 0.02, None, 63)
    else:
        ctx.solve_fixed(np.concatenate([np.arange(8) * 2.0**-10, 7 * 2.0**-10 + np.arange(1, 7) * 2.0**-11]))
    ctx.smooth()
    assert (ctx.get(10) == 0).all(), ctx.get(10)
    out[name] = (ctx.get(11).copy(), ctx.get(12).copy(), ctx.kernel_time_ms(1)[0])
    ctx.close()
a, b = out["persistent"], out["split"]
D = 28 * (q + 1)
diag = np.array([k * (k + 1) // 2 + k for k in range(D)])
sd = np.sqrt(np.maximum(a[1][:, diag], 0)) + 1e-300
dm = np.abs(a[0] - b[0]) / (np.abs(a[0]) + sd)
print("finite:", np.isfinite(b[0]).all() and np.isfinite(b[1]).all())
print("smoothed mean: max |diff| / (|mean| + sd) =", dm.max())
scale = np.abs(a[1]).max(axis=1, keepdims=True) + 1e-300
print("smoothed cov: max |diff| / max |record| =", (np.abs(a[1] - b[1]) / scale).max())
print("time ms persistent / split:", a[2], b[2])
