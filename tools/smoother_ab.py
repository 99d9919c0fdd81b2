#!/usr/bin/env python3
"""A/B of the small-state smoother kernels on one ensemble: the one-lane-per-trajectory kernel (smooth_lane.h) against the
two-lanes-per-trajectory kernel (smooth_pair.h), same filter records.  One JSON line per case: kernel times from the
library's hipEvents (median of --reps launches after a warm-up) and the largest difference between the two results.

usage: smoother_ab.py [--cases lorenz63:3:65536:1024,fhn:3:65536:256] [--reps 3] [--adaptive]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odefilters_jl_amd as pkg
from odefilters_jl_amd.host import F_SMOOTH_MEAN, F_SMOOTH_COV_TRIL, F_RETCODE

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="lorenz63:3:65536:1024")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--adaptive", action="store_true")
ap.add_argument("--compare-steps", type=int, default=64, help="records compared (from the start of the grid; the pass runs backwards, so these are its last)")
a = ap.parse_args()
U0 = {"lorenz63": ([1.0, 0.0, 0.0], [10.0, 28.0, 8.0 / 3.0], 2.0**-9), "fhn": ([-1.0, 1.0], [0.2, 0.2, 3.0], 2.0**-6),
      "lotka_volterra": ([1.0, 1.0], [1.5, 1.0, 3.0, 1.0], 2.0**-7), "vanderpol": ([2.0, 0.0], [1.0], 2.0**-8)}
os.environ["ODEF_SMOOTH_ROWS_MAX_N"] = "0"  # never the row-team kernel here
os.environ["ODEF_SMOOTH_LANE_MIN_N"] = "1"
for case in a.cases.split(","):
    rhs, q, N, ns = case.split(":")
    q, N, ns = int(q), int(N), int(ns)
    u0, p, dt = U0[rhs]
    d = len(u0)
    D = d * (q + 1)
    ctx = pkg.Context(rhs, q, 1, N, smooth=True)
    ctx.set_problem_perturbed(u0, p, 0.0, 1e-2)
    if a.adaptive:
        ctx.solve_adaptive(ns * dt, 1e-6, 1e-3, dt, None, 1024)
    else:
        ctx.solve_fixed(np.arange(ns + 1) * dt)
    out = {"rhs": rhs, "order": q, "D": D, "traj": N, "nsteps": ns, "adaptive": a.adaptive}
    res = {}
    for name, flag in (("lane", "0"), ("pair", "1")):
        os.environ["ODEF_SMOOTH_PAIR"] = flag
        ts = []
        for _ in range(a.reps + 1):
            ctx.smooth()
            ts.append(ctx.kernel_time_ms(1)[0])
        out[name + "_ms"] = float(np.median(ts[1:]))
        k = min(a.compare_steps, ctx.n_save)
        res[name] = (ctx.get(F_SMOOTH_MEAN)[:k].copy(), ctx.get(F_SMOOTH_COV_TRIL)[:k].copy(), ctx.get(F_RETCODE).copy())
    m0, c0, r0 = res["lane"]
    m1, c1, r1 = res["pair"]
    out["retcodes_equal"] = bool((r0 == r1).all())
    out["max_rel_mean_diff_u"] = float(np.nanmax(np.abs(m0[:, :d] - m1[:, :d])) / (np.nanmax(np.abs(m0[:, :d])) + 1e-300))
    out["max_rel_cov_diff"] = float(np.nanmax(np.abs(c0 - c1)) / (np.nanmax(np.abs(c0)) + 1e-300))
    B = 8 * (D + D * (D + 1) // 2 + 1)
    sb = (2 * B - 8) * N * (ns - 1)
    if not a.adaptive:
        out["pair_frac_of_8TBps"] = sb / (out["pair_ms"] * 1e-3) / 8e12
        out["lane_frac_of_8TBps"] = sb / (out["lane_ms"] * 1e-3) / 8e12
    print(json.dumps(out), flush=True)
    ctx.close()
