#!/usr/bin/env python3
"""BASELINE.json configurations 2-5 as the library runs them by default (kernel selection of csrc/ek_kernels.h), one
JSON line each: kernel times from the library's hipEvents (median of `--reps` launches after a warm-up), the
algorithmic bytes / flops of SURVEY.md 8(d) and the roofline fraction they give.  `--only 2,5` restricts the list
(used under rocprofv3, where one configuration per run keeps the traces apart)."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odefilters_jl_amd as pkg

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="2,3s,5,4")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
D, TRI = 12, 78
B = 8 * (D + TRI + 1)
LU0, LP = [1.0, 0.0, 0.0], [10.0, 28.0, 8.0 / 3.0]
dt = 2.0**-9


def med(f, reps):
    f()
    return float(np.median([f() for _ in range(reps)]))


for cfg in a.only.split(","):
    if cfg in ("2", "3s", "3x8"):  # fixed-step Lorenz: filter (every step saved) + smoother
        N = {"2": 4096, "3s": 65536, "3x8": 8192}[cfg]
        ns = 1024
        ctx = pkg.Context("lorenz63", 3, 1, N, smooth=True)
        ctx.set_problem_perturbed(LU0, LP, 0.0, 1e-2)
        tg = np.arange(ns + 1) * dt

        def filt():
            ctx.solve_fixed(tg)
            return ctx.kernel_time_ms(0)[0]

        def smooth():
            ctx.smooth()
            return ctx.kernel_time_ms(1)[0]

        f_ms = med(filt, a.reps)
        s_ms = med(smooth, a.reps)
        fb, sb = B * N * (ns + 1), (2 * B - 8) * N * (ns - 1)
        print(json.dumps({"config": {"2": "2: Lorenz-63 EK1(3), 4 096 x 1 024 steps, every step saved, + RTS smoother",
                                     "3s": "3: Lorenz-63 EK1(3), 65 536 x 1 024 steps, every step saved, + RTS smoother",
                                     "3x8": "3 sharded over 8 GPUs: the 8 192-trajectory shard of one GPU"}[cfg],
                          "traj": N, "nsteps": ns, "filter_ms": f_ms, "smooth_ms": s_ms,
                          "filter_steps_per_s": N * ns / (f_ms * 1e-3), "smoother_steps_per_s": N * (ns - 1) / (s_ms * 1e-3),
                          "filter_roofline": {"bound": "hbm", "achieved": fb / (f_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": fb / (f_ms * 1e-3) / 8e12},
                          "smoother_roofline": {"bound": "hbm", "achieved": sb / (s_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": sb / (s_ms * 1e-3) / 8e12},
                          "retcodes_ok": bool((ctx.get(10) == 0).all())}), flush=True)
        ctx.close()
    elif cfg in ("5", "5x8"):  # adaptive + smoother
        N = {"5": 16384, "5x8": 2048}[cfg]
        ctx = pkg.Context("lorenz63", 3, 1, N, smooth=True)
        ctx.set_problem_perturbed(LU0, LP, 0.0, 1e-2)

        def filt():
            ctx.solve_adaptive(2.0, 1e-6, 1e-3, dt, None, 400)
            return ctx.kernel_time_ms(0)[0]

        def smooth():
            ctx.smooth()
            return ctx.kernel_time_ms(1)[0]

        f_ms = med(filt, a.reps)
        s_ms = med(smooth, a.reps)
        na, nr = ctx.get(5), ctx.get(6)
        att = int(na.sum() + nr.sum())
        print(json.dumps({"config": {"5": "5: Lorenz-63 EK1(3), 16 384 trajectories, adaptive PI + RTS smoother",
                                     "5x8": "5 sharded over 8 GPUs: the 2 048-trajectory shard of one GPU"}[cfg],
                          "traj": N, "filter_ms": f_ms, "smooth_ms": s_ms, "filter_plus_smoother_ms": f_ms + s_ms,
                          "attempted_steps": att, "accepted": int(na.sum()), "attempted_steps_per_s": att / (f_ms * 1e-3),
                          "filter_roofline": {"bound": "hbm", "achieved": (B + 8) * att / (f_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": (B + 8) * att / (f_ms * 1e-3) / 8e12},
                          "retcodes_ok": bool((ctx.get(10) == 0).all())}), flush=True)
        ctx.close()
    elif cfg == "4":
        N, ns = 8192, 256
        u0 = [3.0, 3.0, -1.0, -3.0, 2.0, -2.0, 2.0, 3.0, -3.0, 2.0, 0.0, 0.0, -4.0, 4.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.75, -1.5, 0.0, 0.0, 0.0, -1.25, 1.0, 0.0, 0.0]
        ctx = pkg.Context("pleiades", 5, 1, N, save_everystep=False)
        ctx.set_problem_perturbed(u0, [], 0.0, 1e-3, n_perturbed=14)
        tg = np.arange(ns + 1) * 2.0**-10

        def filt():
            ctx.solve_fixed(tg)
            return ctx.kernel_time_ms(0)[0]

        f_ms = med(filt, max(2, a.reps // 2))
        DD = 168
        F = (16 / 3) * DD**3 + 8 * 28 * DD**2 + 4 * DD**2
        # executed flops of the MFMA kernel per step, counted from the code (tools/bench_modes.py spells the count out)
        mfma = 2 * 546 + 2 * 192 + 132 + 120 + 24 + 24 + 21
        Fx = mfma * 2048 + 0.4e6
        sps = N * ns / (f_ms * 1e-3)
        tiles = os.environ.get("ODEF_PLEIADES_FILTER", "").startswith("t")
        print(json.dumps({"config": "4: Pleiades d=28 EK1(5) (D = 168), 8 192 x 256 steps, final state", "kernel": "tiles (VALU)" if tiles else "mfma",
                          "traj": N, "nsteps": ns, "filter_ms": f_ms, "steps_per_s": sps,
                          "roofline": {"bound": "mfma", "achieved": None if tiles else Fx * sps / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                                       "frac": None if tiles else Fx * sps / 78.6e12, "executed_flop_per_step": None if tiles else Fx,
                                       "F_alg_equivalent_TFLOPs": F * sps / 1e12,
                                       "note": "achieved = executed flops (1 797 v_mfma_f64_16x16x4 per step + ~0.4 Mflop of vector work) x steps/s; the dense-algebra "
                                               "count F_alg of SURVEY 8(d) (31.7 Mflop per step of the reference's square-root form) is 7.7x larger"},
                          "retcodes_ok": bool((ctx.get(10) == 0).all())}), flush=True)
        ctx.close()
