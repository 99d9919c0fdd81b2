"""Diagnostic: device Pleiades filter vs the golden config-4 fixture, per step."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import odefilters_jl_amd as pkg
import importlib
host = importlib.import_module('odefilters.jl_amd.host') if False else sys.modules[[m for m in sys.modules if m.endswith('host') and 'odefilters' in m][0]]
if os.environ.get('ODEF_DIAG_LIB'):
    host.LIB_PATH = os.environ['ODEF_DIAG_LIB']
print('lib', host.LIB_PATH)

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pleiades_ek1_q5_cfg4.npz"))
u0s, ns, dt = g["u0s"], int(g["nsteps"]), float(g["dt"])
np.set_printoptions(precision=4, linewidth=200)
for kind, q in (("EK1", 5),):
    prob = pkg.EnsembleProblem(pkg.ODEProblem("pleiades", u0s[0], (0.0, ns * dt), ()), u0s=u0s)
    sol = pkg.solve(prob, getattr(pkg, kind)(order=q, smooth=False), pkg.EnsembleHIP(), dt=dt, adaptive=False)
    print(kind, q, sol.retcode)
    if kind == "EK1" and q == 5:
        print("u err per step", np.abs(sol.u - g["u"]).max(axis=2))
        print("diff dev", np.asarray(sol.diffusions)[0])
        print("diff gold", g["diffusions"][0])
    else:
        print("u range per step", np.abs(sol.u).max(axis=2)[0])
        print("diff dev", np.asarray(sol.diffusions)[0])
