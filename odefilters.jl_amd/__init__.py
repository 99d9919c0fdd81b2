"""odefilters.jl_amd -- MI355X-native Gaussian ODE-filter hot path (EK0/EK1 predict / update /
smooth of ProbNumDiffEq.jl) behind a C ABI.  Holds only what that path needs:

  csrc/      hand-written HIP kernels for gfx950 + the C-ABI layer (libodefilter_hip.so)
  host.py    host-side mirror of the reference's solver interface (ctypes over the C ABI)
  dist.py    ensemble sharding across GPUs (one process per GPU, one RCCL all-gather)
"""
from . import _build
from .host import (  # noqa: F401
    EK0, EK1, Context, EnsembleHIP, EnsembleProblem, EnsembleSolution, ODEProblem, OdefController, OdefError,
    compile_rhs, fixed_time_grid, ibm, load_library, preconditioner, predict, shard_ensemble, smooth_step, solve, unpack_tril, update,
)

build = _build.build
