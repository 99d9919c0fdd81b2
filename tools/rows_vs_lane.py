#!/usr/bin/env python3
"""Row-team filter (csrc/rows_filter.h, 16 lanes per trajectory) against the lane filter (one lane per trajectory):
kernel time of the Lorenz-63 EK1(3) filter -- fixed step (1 024 steps, every step saved / final state only) and adaptive
(t in [0, 2], abstol 1e-6, reltol 1e-3) -- per ensemble size.  One JSON line per size; the launcher's crossover
(kFilterRowsMaxN, ek_kernels.h) is read off this table (profiles/r02_rows_vs_lane.jsonl)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odefilters_jl_amd as pkg

ns, dt = 1024, 2.0**-9
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384, 32768, 65536]
for N in sizes:
    out = {"traj": N, "nsteps": ns}
    for name, v in (("lane", "0"), ("rows", "1000000000")):
        os.environ["ODEF_FILTER_ROWS_MAX_N"] = v
        os.environ["ODEF_SMOOTH_ROWS_MAX_N"] = v  # lane side: the round-1 choice (LDS row teams below 6 144, lane kernel above)
        for mode, save in (("every", "everystep"), ("final", "final")):
            ctx = pkg.Context("lorenz63", 3, 1, N, save_everystep=(mode == "every"))
            ctx.set_problem_perturbed([1.0, 0.0, 0.0], [10.0, 28.0, 8.0 / 3.0], 0.0, 1e-2)
            ts, tsm = [], []
            for _ in range(4):
                ctx.solve_fixed(np.arange(ns + 1) * dt)
                ts.append(ctx.kernel_time_ms(0)[0])
                if mode == "every":
                    ctx.smooth()
                    tsm.append(ctx.kernel_time_ms(1)[0])
            out[f"{name}_{mode}_ms"] = round(float(np.median(ts[1:])), 4)
            if tsm:
                out[f"{name}_smooth_ms"] = round(float(np.median(tsm[1:])), 4)
            ctx.close()
        ctx = pkg.Context("lorenz63", 3, 1, N, save_everystep=True)
        ctx.set_problem_perturbed([1.0, 0.0, 0.0], [10.0, 28.0, 8.0 / 3.0], 0.0, 1e-2)
        ts, tsm = [], []
        for _ in range(4):
            ctx.solve_adaptive(2.0, 1e-6, 1e-3, dt, max_steps=400)
            ts.append(ctx.kernel_time_ms(0)[0])
            ctx.smooth()
            tsm.append(ctx.kernel_time_ms(1)[0])
        out[f"{name}_adaptive_ms"] = round(float(np.median(ts[1:])), 4)
        out[f"{name}_adaptive_smooth_ms"] = round(float(np.median(tsm[1:])), 4)
        ctx.close()
    print(json.dumps(out), flush=True)
