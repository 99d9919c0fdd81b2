#!/usr/bin/env python3
"""Row-team filter (csrc/filter_rows.h) against the lane filter on small ensembles: kernel time of the fixed-step
Lorenz-63 EK1(3) filter, every step saved, 1 024 steps.  One JSON line per ensemble size."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odefilters_jl_amd as pkg

ns, dt = 1024, 2.0**-9
for N in (256, 1024, 2048, 4096, 8192, 16384):
    out = {"traj": N, "nsteps": ns}
    for name, v in (("lane_ms", "0"), ("rows_ms", "1000000000")):
        os.environ["ODEF_FILTER_ROWS_MAX_N"] = v
        ctx = pkg.Context("lorenz63", 3, 1, N, smooth=False)
        ctx.set_problem_perturbed([1.0, 0.0, 0.0], [10.0, 28.0, 8.0 / 3.0], 0.0, 1e-2)
        for _ in range(3):
            ctx.solve_fixed(np.arange(ns + 1) * dt)
        out[name] = ctx.kernel_time_ms(0)[0]
        ctx.close()
    print(json.dumps(out))
