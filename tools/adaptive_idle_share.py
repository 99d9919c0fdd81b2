#!/usr/bin/env python3
"""How much of an adaptive solve do the lanes / teams of a group spend waiting for the slowest trajectory of the group?
BASELINE config 5 (Lorenz-63 EK1(3), 16 384 trajectories, abstol 1e-6, reltol 1e-3): attempts per trajectory, and for groups of
16 (a row-team workgroup), 64 (a wavefront of the lane kernel) and 256 trajectories the share 1 - mean / mean-of-group-maxima.
DESIGN.md section 7, item 5 (the device-side ticket of SURVEY.md:324 would recover at most this share)."""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import odefilters_jl_amd as pkg
N=16384
ctx = pkg.Context("lorenz63", 3, 1, N, save_everystep=True)
ctx.set_problem_perturbed([1.0,0.0,0.0],[10.0,28.0,8.0/3.0],0.0,1e-2)
ctx.solve_adaptive(2.0,1e-6,1e-3,2.0**-9,max_steps=400)
ns=ctx.get(9).astype(float)-1   # attempts per trajectory
for g in (16,64,256):
    m=ns.reshape(-1,g)
    print(f"group of {g}: attempts mean {ns.mean():.1f}, mean of group maxima {m.max(1).mean():.1f}, idle share {1-ns.mean()/m.max(1).mean():.3f}, global max {ns.max():.0f} min {ns.min():.0f}")
