"""ctypes driver for tests/emul/libodef_emul.so (host build of the per-lane device source).
Test infrastructure only."""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import odefilter_oracle as orc  # noqa: E402

MAXNB = 6
_LIB = None


def build():
    src = os.path.join(HERE, "emul", "emul.cpp")
    out = os.path.join(HERE, "emul", "libodef_emul.so")
    import glob

    deps = [src] + glob.glob(os.path.join(ROOT, "odefilters.jl_amd", "csrc", "*.h"))  # every device header
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++20", "-shared", "-fPIC", "-Wno-unknown-pragmas", src, "-o", out])
    return out


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class EmulArgs(C.Structure):
    _fields_ = [
        ("rhs", C.c_int), ("q", C.c_int), ("ek1", C.c_int), ("adaptive", C.c_int),
        ("N", C.c_long), ("u0", dp), ("p", dp), ("p_shared", C.c_int),
        ("At", dp), ("Qt", dp), ("QLt", dp),
        ("hs", dp), ("ptab", dp), ("tab_idx", ip), ("nsteps", C.c_long),
        ("t0", C.c_double), ("t1", C.c_double), ("abstol", C.c_double), ("reltol", C.c_double), ("dt0", C.c_double),
        ("ctrl", dp), ("max_save", C.c_long),
        ("everystep", C.c_int), ("fixed_diffusion", C.c_int), ("want_loglik", C.c_int),
        ("mean", dp), ("cov", dp), ("diff", dp), ("tsave", dp), ("loglik", dp),
        ("naccept", ip), ("nreject", ip), ("nf", ip), ("njac", ip), ("nsaved", ip), ("retcode", ip),
        ("smean", dp), ("scov", dp), ("n_save", C.c_long),
    ]


def prior_tables(q):
    """(q+1)x(q+1) scalar blocks of A, Q, Q_L (priors.jl:7-59) padded to MAXNB."""
    A, QL = orc.ibm(1, q)
    Q = np.zeros((q + 1, q + 1))
    for r in range(q + 1):
        for c in range(q + 1):
            Q[r, c] = 1.0 / ((2 * q + 1 - r - c) * math.factorial(q - r) * math.factorial(q - c))
    out = []
    for M in (A, Q, QL):
        T = np.zeros((MAXNB, MAXNB))
        T[: q + 1, : q + 1] = M
        out.append(np.ascontiguousarray(T))
    return out


def _p(a, t=dp):
    return a.ctypes.data_as(t)


def unpack_tril(c, D):
    """[..., TRI] -> [..., D, D] symmetric."""
    out = np.zeros(c.shape[:-1] + (D, D))
    k = 0
    for i in range(D):
        for j in range(i + 1):
            out[..., i, j] = c[..., k]
            out[..., j, i] = c[..., k]
            k += 1
    return out


def emul_solve(rhs_id, d, q, ek1, u0s, p, *, team=False, tgrid=None, adaptive=False, t0=0.0, t1=1.0, abstol=1e-6, reltol=1e-3,
               dt0=1e-2, max_save=4096, everystep=True, fixed_diffusion=False, want_loglik=True, smooth=False,
               ctrl=None, dense_t=None, sample=None, dense_sample=None):
    """u0s [N, d]; p [np] shared.  Returns dict of numpy arrays in the device layout transposed
    to trajectory-major: mean [N, n_save, D], cov [N, n_save, D, D] ..."""
    u0s = np.asarray(u0s, float)
    N = u0s.shape[0]
    D = d * (q + 1)
    TRI = D * (D + 1) // 2
    At, Qt, QLt = prior_tables(q)
    u0_dev = np.ascontiguousarray(u0s.T)
    p = np.ascontiguousarray(np.asarray(p, float)) if len(p) else np.zeros(1)
    if ctrl is None:
        ctrl = np.array([7.0 / (10 * (q + 1)), 2.0 / (5 * (q + 1)), 0.9, 0.2, 10.0, 1.0, 1.0, 1e-4, 0.0, 1e300])
    a = EmulArgs()
    a.rhs, a.q, a.ek1, a.adaptive = rhs_id, q, int(ek1), int(adaptive)
    a.N, a.u0, a.p, a.p_shared = N, _p(u0_dev), _p(p), 1
    a.At, a.Qt, a.QLt = _p(At), _p(Qt), _p(QLt)
    if adaptive:
        n_save = max_save
        hs = ptab = tg = np.zeros(1)
        tab_idx = np.zeros(1, np.int32)
        nsteps = 0
    else:
        tg = np.ascontiguousarray(np.asarray(tgrid, float))
        hs = np.ascontiguousarray(np.diff(tg))
        nsteps = len(hs)
        n_save = nsteps + 1 if int(everystep) > 0 else 1
        uniq, inv = np.unique(hs, return_inverse=True)
        stride = lib().emul_tab_stride()
        ptab = np.zeros((len(uniq), stride))
        lib().emul_precond_fill.argtypes = [C.c_int, C.c_double, C.c_double, dp]
        for k, h in enumerate(uniq):
            lib().emul_precond_fill(q, float(h), float(h) ** (-q - 1 / 2), _p(ptab[k]))
        tab_idx = np.ascontiguousarray(inv.astype(np.int32))
    a.hs, a.ptab, a.tab_idx, a.nsteps = _p(hs), _p(ptab), _p(tab_idx, ip), nsteps
    a.t0, a.t1, a.abstol, a.reltol, a.dt0 = t0, t1, abstol, reltol, dt0
    a.ctrl, a.max_save = _p(ctrl), max_save
    a.everystep, a.fixed_diffusion, a.want_loglik = int(everystep), int(fixed_diffusion), int(want_loglik)
    mean = np.zeros((n_save, D, N)); cov = np.zeros((n_save, TRI, N)); diff = np.zeros((n_save, N))
    tsave = np.zeros((n_save, N)); loglik = np.zeros(N)
    ints = [np.zeros(N, np.int32) for _ in range(6)]
    a.mean, a.cov, a.diff, a.tsave, a.loglik = _p(mean), _p(cov), _p(diff), _p(tsave), _p(loglik)
    a.naccept, a.nreject, a.nf, a.njac, a.nsaved, a.retcode = [_p(x, ip) for x in ints]
    smean = np.zeros_like(mean) if smooth else np.zeros(1)
    scov = np.zeros_like(cov) if smooth else np.zeros(1)
    a.smean, a.scov, a.n_save = _p(smean), _p(scov), n_save
    fn = lib().emul_filter_tiles if team == "tiles" else (lib().emul_filter_team if team else lib().emul_filter)
    rc = fn(C.byref(a))
    assert rc == 0, rc
    out = dict(mean=mean.transpose(2, 0, 1), cov=unpack_tril(cov.transpose(2, 0, 1), D), diff=diff.T, tsave=tsave.T,
               loglik=loglik, naccept=ints[0], nreject=ints[1], nf=ints[2], njac=ints[3], nsaved=ints[4],
               retcode=ints[5], tgrid=tg)
    if smooth:
        rc = lib().emul_smooth(C.byref(a), d)
        assert rc == 0, rc
        out["smean"] = smean.transpose(2, 0, 1)
        out["scov"] = unpack_tril(scov.transpose(2, 0, 1), D)
    if sample is not None:  # (n_samples, seed, noise_scale)
        class EmulSample(C.Structure):
            _fields_ = [("a", C.POINTER(EmulArgs)), ("n_samples", C.c_long), ("seed", C.c_ulonglong), ("noise_scale", C.c_double), ("samples", dp)]
        ns_, seed_, scale_ = sample
        smp = np.zeros((n_save, D, ns_, N))
        e = EmulSample(C.pointer(a), ns_, seed_, scale_, _p(smp))
        rc = lib().emul_sample(C.byref(e), d)
        assert rc == 0, rc
        out["samples"] = smp.transpose(3, 0, 1, 2)  # [N, n_save, D, n_samples]
    if adaptive:
        # one record per ATTEMPTED step on the device side; drop the repeats of rejected attempts (unchanged time),
        # as host.py EnsembleSolution does, so that the arrays line up with the reference's accepted-only records
        ns_raw = ints[4].copy()
        out["nsaved_raw"] = ns_raw
        tt = tsave.T
        keep = np.arange(n_save)[None, :] < ns_raw[:, None]
        keep[:, 1:] &= tt[:, 1:] != tt[:, :-1]
        order = np.argsort(~keep, axis=1, kind="stable")
        count = keep.sum(axis=1).astype(np.int32)
        out["raw_index"] = [order[i, : count[i]] for i in range(N)]
        out["nsaved"] = count
        for key in ("mean", "cov", "diff", "tsave", "smean", "scov", "samples"):
            if key in out:
                arr = out[key]
                c = np.take_along_axis(arr, order.reshape(order.shape + (1,) * (arr.ndim - 2)), axis=1)
                c[np.arange(n_save)[None, :] >= count[:, None]] = 0
                out[key] = c
    if dense_t is not None:
        class EmulDense(C.Structure):
            _fields_ = [("a", C.POINTER(EmulArgs)), ("smoothed", C.c_int), ("tq", dp), ("n_q", C.c_long), ("qmean", dp), ("qcov", dp)]
        tq = np.ascontiguousarray(dense_t, float)
        qm = np.zeros((len(tq), D, N)); qc = np.zeros((len(tq), TRI, N))
        if not adaptive:
            a.hs = _p(tg)  # emul_dense reads the fixed time grid from here
        e = EmulDense(C.pointer(a), int(smooth), _p(tq), len(tq), _p(qm), _p(qc))
        rc = lib().emul_dense(C.byref(e), d)
        assert rc == 0, rc
        out["qmean"] = qm.transpose(2, 0, 1)
        out["qcov"] = unpack_tril(qc.transpose(2, 0, 1), D)
    if dense_sample is not None:  # (times, n_samples, seed, noise_scale): filter posterior at the times, then the sampler
        class EmulDense(C.Structure):
            _fields_ = [("a", C.POINTER(EmulArgs)), ("smoothed", C.c_int), ("tq", dp), ("n_q", C.c_long), ("qmean", dp), ("qcov", dp)]

        class EmulDenseSample(C.Structure):
            _fields_ = [("a", C.POINTER(EmulArgs)), ("tq", dp), ("n_q", C.c_long), ("qmean", dp), ("qcov", dp), ("rec_t", dp),
                        ("n_samples", C.c_long), ("seed", C.c_ulonglong), ("noise_scale", C.c_double), ("samples", dp)]
        times, ns_, seed_, scale_ = dense_sample
        tq = np.ascontiguousarray(times, float)
        qm = np.zeros((len(tq), D, N)); qc = np.zeros((len(tq), TRI, N))
        if not adaptive:
            a.hs = _p(tg)
        e = EmulDense(C.pointer(a), 0, _p(tq), len(tq), _p(qm), _p(qc))
        rc = lib().emul_dense(C.byref(e), d)
        assert rc == 0, rc
        smp = np.zeros((len(tq), D, ns_, N))
        e2 = EmulDenseSample(C.pointer(a), _p(tq), len(tq), _p(qm), _p(qc), _p(tg), ns_, seed_, scale_, _p(smp))
        rc = lib().emul_dense_sample(C.byref(e2), d)
        assert rc == 0, rc
        out["dense_samples"] = smp.transpose(3, 0, 1, 2)  # [N, n_q, D, n_samples]
    return out
