"""ctypes wrapper of oracle/libodefilter_cport.so (C restatement of the reference's fixed-step
filter loop, OpenMP over trajectories).  TEST INFRASTRUCTURE / CPU BASELINE ONLY."""
import ctypes as C
import os
import time

import numpy as np

import odefilter_oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libodefilter_cport.so")
_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        dp = C.POINTER(C.c_double)
        _lib.cport_filter_fixed.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, dp, dp, dp, dp, dp, C.c_long, dp, dp, dp, C.c_int]
        _lib.cport_filter_fixed.restype = C.c_int
        _lib.cport_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def filter_fixed(rhs: str, q: int, ek1: bool, u0s: np.ndarray, p: np.ndarray, tgrid: np.ndarray, nthreads: int = 0):
    """Final (mean [N, D], cov [N, D, D]) of the fixed-step filter for each initial value."""
    vf = orc.vector_field(rhs)
    rhs_id = {"fhn": 0, "lorenz63": 1, "lotka_volterra": 2}[rhs]
    d = vf.d
    D = d * (q + 1)
    A, QL = orc.ibm(d, q)
    A, QL = np.ascontiguousarray(A), np.ascontiguousarray(QL)
    hs = np.ascontiguousarray(np.diff(tgrid))
    pv = np.array([h ** (-q - 1 / 2) for h in hs])
    m0 = np.ascontiguousarray([orc.initial_update(u0, vf, p, tgrid[0], q).mu for u0 in u0s])
    N = len(u0s)
    mean, cov = np.zeros((N, D)), np.zeros((N, D, D))
    pp = np.ascontiguousarray(p, float)
    t0 = time.perf_counter()
    rc = lib().cport_filter_fixed(rhs_id, d, q, int(ek1), N, _p(A), _p(QL), _p(pp), _p(hs), _p(pv), len(hs), _p(m0), _p(mean), _p(cov), nthreads)
    el = time.perf_counter() - t0
    assert rc >= 0
    return mean, cov, el


def effective_cores() -> int:
    """Host cores this process may actually use: CPU affinity capped by the cgroup CPU quota
    (a container can see 128 logical CPUs and be granted 16 of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def bench_lorenz(seconds_budget: float = 15.0):
    """cpu_baseline leg of bench.py: Lorenz-63 EK1(3), dt = 2^-9, all host cores granted to this process."""
    vf = orc.vector_field("lorenz63")
    threads = max(1, min(lib().cport_max_threads(), effective_cores()))
    nsteps = 1024
    tg = np.arange(nsteps + 1) * 2.0**-9
    n = 16 * threads
    u0s = orc.ensemble_u0(vf.u0, n, 1e-2)
    _, _, el = filter_fixed("lorenz63", 3, True, u0s, vf.p, tg, threads)  # calibration
    n2 = int(max(n, min(65536, n * seconds_budget / max(el, 1e-6) // threads * threads)))
    u0s = orc.ensemble_u0(vf.u0, min(n2, 4096), 1e-2)
    reps = max(1, n2 // len(u0s))
    best = None
    tot = 0.0
    for _ in range(min(reps, 3)):
        _, _, el = filter_fixed("lorenz63", 3, True, u0s, vf.p, tg, threads)
        tot += el
        best = el if best is None else min(best, el)
    return {"value": len(u0s) * nsteps / best, "unit": "filter steps/s", "cores": threads, "kind": "port",
            "sample": f"{len(u0s)} trajectories x {nsteps} steps of the same Lorenz-63 EK1(3) workload, C restatement "
                      f"(oracle/odefilter_cport.c, -O3 -march=native, OpenMP x{threads}), best of {min(reps, 3)}"}
