#!/usr/bin/env python3
"""D = 168 filter kernels side by side: the MFMA kernel (csrc/filter_mfma.h, default) against the register-tiled VALU
kernel (csrc/filter_tiles.h, ODEF_PLEIADES_FILTER=tiles) on the same inputs: per saved step, the largest relative
difference of the mean per derivative block and of the covariance.  Debugging aid / cross-check (both are compared
with the oracle in tests/)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odefilters_jl_amd as pkg
from odefilters_jl_amd.host import unpack_tril

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2
order = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N, dt, d = 2, 2.0**-10, 28
out = {}
base = np.array([3, 3, -1, -3, 2, -2, 2, 3, -3, 2, 0, 0, -4, 4, 0, 0, 0, 0, 0, 1.75, -1.5, 0, 0, 0, -1.25, 1, 0, 0], float)
for name, env in (("mfma", ""), ("tiles", "tiles")):
    os.environ["ODEF_PLEIADES_FILTER"] = env
    ctx = pkg.Context("pleiades", order, 1, N, save_everystep=True)
    ctx.set_problem_perturbed(base, [], 0.0, 1e-3, n_perturbed=14)
    ctx.solve_fixed(np.arange(ns + 1) * dt)
    D = ctx.D
    m = ctx.get(0)[:, :, 0]                       # [n_save][D]
    c = np.stack([unpack_tril(ctx.get(1)[s, :, 0], D) for s in range(ns + 1)])
    out[name] = (m, c, np.array(ctx.get(10)), ctx.get(2)[:, 0])
    ctx.close()
m1, c1, r1, d1 = out["mfma"]
m0, c0, r0, d0 = out["tiles"]
print("retcodes mfma", r1, "tiles", r0, "shapes", m1.shape, c1.shape)
if d1 is not None:
    print("diffusions mfma", d1[:6], "tiles", d0[:6])
D = m1.shape[-1]
for s in sorted(set(list(range(min(ns + 1, 4))) + [int(x) for x in np.linspace(0, ns, 14)])):
    a, b = m1[s], m0[s]
    be = [np.max(np.abs(a[k * d:(k + 1) * d] - b[k * d:(k + 1) * d])) / (np.max(np.abs(b[k * d:(k + 1) * d])) + 1e-300) for k in range(D // d)]
    ca, cb = c1[s], c0[s]
    ce = np.max(np.abs(ca - cb)) / (np.max(np.abs(cb)) + 1e-300)
    nan = np.isnan(a).sum(), np.isnan(ca).sum()
    print(f"step {s}: mean blocks {np.array2string(np.array(be), precision=2)} cov {ce:.2e} nan {nan}")
    if s >= 1 and ce > 1e-6 and not np.isnan(ce):
        diff = np.abs(ca - cb) / (np.max(np.abs(cb)) + 1e-300)
        bad = np.argwhere(diff > 1e-6)
        print("   first bad entries", bad[:10].tolist(), "count", len(bad))
