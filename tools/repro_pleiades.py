import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import odefilters_jl_amd as pkg
from oracle import odefilter_oracle as orc
mode = sys.argv[1]; q = int(sys.argv[2]); N = int(sys.argv[3]); ns = int(sys.argv[4])
vf = orc.vector_field("pleiades")
ctx = pkg.Context("pleiades", q, 1, N, save_everystep=True)
ctx.set_problem_perturbed(vf.u0, [], 0.0, 1e-3, n_perturbed=14)
ctx.solve_fixed(np.arange(ns + 1) * 2.0**-10)
print("filter ok", np.isfinite(ctx.get(1)).all(), flush=True)
if mode != "filter":
    ctx.smooth()
    print("smooth ok", np.isfinite(ctx.get(12)).all(), flush=True)
ctx.close()
