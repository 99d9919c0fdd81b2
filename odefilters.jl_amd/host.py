"""Host-side mirror of the reference's solver interface for the ensemble hot path.

The reference is Julia (`solve(prob, EK1(order=3); adaptive, dt, abstol, reltol)`,
src/algorithms.jl:23-51, src/solution.jl:8-24); there is no Julia toolchain in this image,
so the host side above the C ABI (include/odefilter.h) is written in Python with the same
names, keyword meanings and error behaviour.  `julia/ODEFilterHIP.jl` is the `ccall`
binding a maintainer of the reference would add.

Everything numerical happens in libodefilter_hip.so (HIP kernels for gfx950).  There is no
CPU path: if the library is missing or no GPU is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ODEFILTER_HIP_LIB: development override to A/B-test another build of the same library
LIB_PATH = os.environ.get("ODEFILTER_HIP_LIB") or os.path.join(_HERE, "lib", "libodefilter_hip.so")

# ---- enums (include/odefilter.h) ---------------------------------------------------------
EK0_ID, EK1_ID = 0, 1
DIFFUSION = {"dynamic": 0, "fixed": 1, "fixedMAP": 2}
RHS = {"fhn": 0, "lorenz63": 1, "lotka_volterra": 2, "vanderpol": 3, "linear": 4, "pleiades": 5, "lorenz96": 6}
RHS_DIMS = {"fhn": (2, 3), "lorenz63": (3, 3), "lotka_volterra": (2, 4), "vanderpol": (2, 1), "linear": (2, 2),
            "pleiades": (28, 0), "lorenz96": (16, 1)}
SAVE_FINAL, SAVE_EVERYSTEP = 0, 1
RETCODES = {0: "Success", 1: "MaxIters", 2: "DtLessThanMin", 3: "Unstable", 4: "Unstable"}
(F_MEAN, F_COV_TRIL, F_DIFFUSION, F_T, F_LOGLIK, F_NACCEPT, F_NREJECT, F_NF, F_NJAC, F_NSAVED, F_RETCODE,
 F_SMOOTH_MEAN, F_SMOOTH_COV_TRIL, F_U0, F_DENSE_MEAN, F_DENSE_COV_TRIL, F_SAMPLES) = range(17)
_INT_FIELDS = {F_NACCEPT, F_NREJECT, F_NF, F_NJAC, F_NSAVED, F_RETCODE}
MAX_ORDER = 5


class OdefConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("alg", C.c_int32), ("order", C.c_int32), ("diffusion", C.c_int32),
        ("smooth", C.c_int32), ("rhs_id", C.c_int32), ("d", C.c_int32), ("n_params", C.c_int32),
        ("params_shared", C.c_int32), ("save_mode", C.c_int32), ("device", C.c_int32), ("want_loglik", C.c_int32),
        ("n_traj", C.c_int64),
    ]


class OdefController(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("beta1", "beta2", "gamma", "qmin", "qmax", "qsteady_min", "qsteady_max", "qoldinit", "dtmin", "dtmax")]


# every symbol include/odefilter.h declares: name -> (restype, argtypes)
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
SYMBOLS = {
    "odef_version": (C.c_int, []),
    "odef_last_error": (C.c_char_p, [_vp]),
    "odef_create": (C.c_int, [C.POINTER(_vp), C.POINTER(OdefConfig)]),
    "odef_rhs_compile": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.POINTER(C.c_int32)]),
    "odef_destroy": (None, [_vp]),
    "odef_set_stream": (C.c_int, [_vp, _vp]),
    "odef_set_problem": (C.c_int, [_vp, _dp, _dp, C.c_double]),
    "odef_set_problem_device": (C.c_int, [_vp, _vp, _vp, C.c_double]),
    "odef_set_problem_perturbed": (C.c_int, [_vp, _dp, _dp, C.c_double, C.c_double, C.c_uint64, C.c_int64, C.c_int32]),
    "odef_solve_fixed": (C.c_int, [_vp, _dp, C.c_int64]),
    "odef_solve_adaptive": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(OdefController), C.c_int64]),
    "odef_smooth": (C.c_int, [_vp]),
    "odef_dense_output": (C.c_int, [_vp, _dp, C.c_int64, C.c_int]),
    "odef_sample": (C.c_int, [_vp, C.c_int64, C.c_uint64, C.c_double]),
    "odef_dense_sample": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.c_uint64, C.c_double]),
    "odef_n_save": (C.c_int64, [_vp]),
    "odef_field_bytes": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_size_t)]),
    "odef_get": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t]),
    "odef_get_device": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "odef_bind_device": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t]),
    "odef_synchronize": (C.c_int, [_vp]),
    "odef_kernel_time_ms": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "odef_kernel_name": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_size_t]),
    "odef_ibm": (C.c_int, [C.c_int, C.c_int, _dp, _dp]),
    "odef_preconditioner": (C.c_int, [C.c_int, C.c_int, C.c_double, _dp]),
    "odef_predict": (C.c_int, [C.c_int, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp]),
    "odef_update": (C.c_int, [C.c_int, C.c_int, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp]),
    "odef_smooth_step": (C.c_int, [C.c_int, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    # ensemble sharded over the GPUs of one node by one process
    "odef_shard_range": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "odef_group_layout": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "odef_unpad_gathered": (C.c_int, [_dp, C.c_int32, C.c_int32, C.c_int64, _dp]),
    "odef_group_create": (C.c_int, [C.POINTER(_vp), C.POINTER(OdefConfig), C.c_int32, C.POINTER(C.c_int32)]),
    "odef_group_destroy": (None, [_vp]),
    "odef_group_last_error": (C.c_char_p, [_vp]),
    "odef_group_size": (C.c_int32, [_vp]),
    "odef_group_ctx": (_vp, [_vp, C.c_int32]),
    "odef_group_shard": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "odef_group_set_problem": (C.c_int, [_vp, _dp, _dp, C.c_double]),
    "odef_group_set_problem_perturbed": (C.c_int, [_vp, _dp, _dp, C.c_double, C.c_double, C.c_uint64, C.c_int32]),
    "odef_group_solve_fixed": (C.c_int, [_vp, _dp, C.c_int64]),
    "odef_group_solve_adaptive": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(OdefController), C.c_int64]),
    "odef_group_smooth": (C.c_int, [_vp]),
    "odef_allgather": (C.c_int, [_vp, C.c_int]),
    "odef_group_get_gathered": (C.c_int, [_vp, C.c_int32, _dp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
}

_lib = None


def load_library() -> C.CDLL:
    """Load libodefilter_hip.so; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C odefilters.jl_amd/csrc).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class OdefError(RuntimeError):
    pass


def shard_range(n_traj: int, n_shards: int, shard: int):
    """(first, count) of a shard: contiguous blocks, the first n_traj % n_shards one longer (`odef_shard_range`; host
    arithmetic only, works without a GPU)."""
    first, count = C.c_int64(), C.c_int64()
    if load_library().odef_shard_range(int(n_traj), int(n_shards), int(shard), C.byref(first), C.byref(count)) != 0:
        raise OdefError(f"odef_shard_range({n_traj}, {n_shards}, {shard}): invalid arguments")
    return int(first.value), int(count.value)


class DeviceGroup:
    """`odef_group`: one ensemble sharded over the GPUs of a node by THIS process (what a Julia host binds with ccall,
    julia/ODEFilterHIP.jl) -- contiguous blocks, no exchange while stepping, `allgather()` = the one RCCL collective."""

    def __init__(self, rhs: str, order: int, alg: int, n_traj: int, n_devices: int, *, device_ids=None, diffusion="dynamic",
                 smooth=False, save_everystep=True, want_loglik=True):
        self.lib = load_library()
        d, npar = RHS_DIMS[rhs]
        cfg = OdefConfig()
        cfg.struct_size = C.sizeof(OdefConfig)
        cfg.alg, cfg.order, cfg.diffusion = alg, order, DIFFUSION[diffusion]
        cfg.smooth, cfg.rhs_id, cfg.d, cfg.n_params = int(smooth), RHS[rhs], d, npar
        cfg.params_shared = 1
        cfg.save_mode = SAVE_EVERYSTEP if (save_everystep or smooth) else SAVE_FINAL
        cfg.device, cfg.want_loglik, cfg.n_traj = -1, int(want_loglik), n_traj
        self.d, self.D, self.N, self.G = d, d * (order + 1), n_traj, n_devices
        ids = (C.c_int32 * n_devices)(*device_ids) if device_ids is not None else None
        h = _vp()
        if self.lib.odef_group_create(C.byref(h), C.byref(cfg), n_devices, ids) != 0:
            raise OdefError(self.lib.odef_group_last_error(None).decode())
        self._h = h

    def _chk(self, rc):
        if rc != 0:
            raise OdefError(self.lib.odef_group_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.odef_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def shard(self, g: int):
        first, count = C.c_int64(), C.c_int64()
        self._chk(self.lib.odef_group_shard(self._h, g, C.byref(first), C.byref(count)))
        return int(first.value), int(count.value)

    def set_problem(self, u0, p, t0):
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        pp = _as_dp(np.ascontiguousarray(p, dtype=np.float64)) if p is not None and len(p) else None
        self._chk(self.lib.odef_group_set_problem(self._h, _as_dp(u0), pp, float(t0)))

    def set_problem_perturbed(self, base_u0, p, t0, scale, seed=0x0DEF17E5, n_perturbed=None):
        base = np.ascontiguousarray(base_u0, dtype=np.float64)
        pp = _as_dp(np.ascontiguousarray(p, dtype=np.float64)) if p is not None and len(p) else None
        self._chk(self.lib.odef_group_set_problem_perturbed(self._h, _as_dp(base), pp, float(t0), float(scale), C.c_uint64(seed),
                                                            self.d if n_perturbed is None else n_perturbed))

    def solve_fixed(self, tgrid):
        tg = np.ascontiguousarray(tgrid, dtype=np.float64)
        self._chk(self.lib.odef_group_solve_fixed(self._h, _as_dp(tg), len(tg)))

    def solve_adaptive(self, t1, abstol=1e-6, reltol=1e-3, dt0=1e-3, max_steps=4096):
        self._chk(self.lib.odef_group_solve_adaptive(self._h, float(t1), float(abstol), float(reltol), float(dt0), None, int(max_steps)))

    def smooth(self):
        self._chk(self.lib.odef_group_smooth(self._h))

    def allgather(self, smoothed=False, from_device=0) -> np.ndarray:
        """Final posterior means of the WHOLE ensemble, [D, N], as device `from_device` holds them after the all-gather."""
        self._chk(self.lib.odef_allgather(self._h, int(smoothed)))
        out = np.empty((self.D, self.N))
        self._chk(self.lib.odef_group_get_gathered(self._h, from_device, _as_dp(out), None, None))
        return out

    def shard_kernel_ms(self, which=0):
        ms = []
        for g in range(self.G):
            t, n = C.c_float(), C.c_int()
            self.lib.odef_kernel_time_ms(self.lib.odef_group_ctx(self._h, g), which, C.byref(t), C.byref(n))
            ms.append(float(t.value))
        return ms


def compile_rhs(name: str, source: str, d: int, n_params: int, struct_name: Optional[str] = None) -> str:
    """Register a user vector field from HIP C++ source (`odef_rhs_compile`, include/odefilter.h): the stand-in for the
    closure `f` (+ `jac`) the reference calls at src/perform_step.jl:106,116-121.  `source` defines a struct
    `struct_name` (default: `name`) with the interface of csrc/rhs.h.  Afterwards `name` can be used wherever a
    compiled-in vector field name is accepted (`ODEProblem(name, ...)`, `Context(name, ...)`).  Raises OdefError with
    the compiler log when the text does not compile."""
    lib = load_library()
    rid = C.c_int32(-1)
    inc = os.path.join(_HERE, "csrc").encode()
    rc = lib.odef_rhs_compile((struct_name or name).encode(), source.encode(), int(d), int(n_params), inc, C.byref(rid))
    if rc != 0:
        raise OdefError(lib.odef_last_error(None).decode())
    RHS[name] = int(rid.value)
    RHS_DIMS[name] = (int(d), int(n_params))
    return name


def _as_dp(a: np.ndarray):
    return a.ctypes.data_as(_dp)


class Context:
    """RAII wrapper of `odef_ctx` (the device-side `GaussianODEFilterCache`, src/caches.jl:5-40)."""

    def __init__(self, rhs: str, order: int, alg: int, n_traj: int, *, diffusion="dynamic", smooth=False,
                 save_everystep=True, params_shared=True, device=-1, want_loglik=True):
        self.lib = load_library()
        if rhs not in RHS:
            raise OdefError(f"unknown vector field {rhs!r}; the device registry has {sorted(RHS)}")
        d, npar = RHS_DIMS[rhs]
        cfg = OdefConfig()
        cfg.struct_size = C.sizeof(OdefConfig)
        cfg.alg, cfg.order, cfg.diffusion = alg, order, DIFFUSION[diffusion]
        cfg.smooth, cfg.rhs_id, cfg.d, cfg.n_params = int(smooth), RHS[rhs], d, npar
        cfg.params_shared = int(params_shared)
        cfg.save_mode = SAVE_EVERYSTEP if (save_everystep or smooth) else SAVE_FINAL
        cfg.device, cfg.want_loglik, cfg.n_traj = device, int(want_loglik), n_traj
        self.cfg = cfg
        self.d, self.q, self.N = d, order, n_traj
        self.D = d * (order + 1)
        self.TRI = self.D * (self.D + 1) // 2
        h = _vp()
        rc = self.lib.odef_create(C.byref(h), C.byref(cfg))
        if rc != 0:
            raise OdefError(self.lib.odef_last_error(None).decode())
        self._h = h

    def _chk(self, rc):
        if rc != 0:
            raise OdefError(self.lib.odef_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.odef_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_ptr: int):
        self._chk(self.lib.odef_set_stream(self._h, _vp(stream_ptr)))

    def set_problem(self, u0: np.ndarray, p: Optional[np.ndarray], t0: float):
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        if u0.shape != (self.N, self.d):
            raise OdefError(f"u0 must have shape ({self.N}, {self.d}), got {u0.shape}")
        pp = None
        if self.cfg.n_params > 0:
            p = np.ascontiguousarray(p, dtype=np.float64)
            want = (self.cfg.n_params,) if self.cfg.params_shared else (self.N, self.cfg.n_params)
            if p.shape != want:
                raise OdefError(f"p must have shape {want}, got {p.shape}")
            pp = _as_dp(p)
        self._chk(self.lib.odef_set_problem(self._h, _as_dp(u0), pp, float(t0)))

    def set_problem_device(self, d_u0_ptr: int, d_p_ptr: int, t0: float):
        self._chk(self.lib.odef_set_problem_device(self._h, _vp(d_u0_ptr), _vp(d_p_ptr), float(t0)))

    def set_problem_perturbed(self, base_u0, p, t0, scale, seed=0x0DEF17E5, first_index=0, n_perturbed=None):
        base = np.ascontiguousarray(base_u0, dtype=np.float64)
        pp = _as_dp(np.ascontiguousarray(p, dtype=np.float64)) if self.cfg.n_params > 0 else None
        npert = self.d if n_perturbed is None else n_perturbed
        self._chk(self.lib.odef_set_problem_perturbed(self._h, _as_dp(base), pp, float(t0), float(scale),
                                                      C.c_uint64(seed), C.c_int64(first_index), npert))

    def solve_fixed(self, tgrid: Sequence[float]):
        tg = np.ascontiguousarray(tgrid, dtype=np.float64)
        self._chk(self.lib.odef_solve_fixed(self._h, _as_dp(tg), len(tg)))

    def solve_adaptive(self, t1, abstol=1e-6, reltol=1e-3, dt0=1e-3, controller: Optional[OdefController] = None,
                       max_steps=4096):
        cp = C.byref(controller) if controller is not None else None
        self._chk(self.lib.odef_solve_adaptive(self._h, float(t1), float(abstol), float(reltol), float(dt0), cp,
                                               int(max_steps)))

    def smooth(self):
        self._chk(self.lib.odef_smooth(self._h))

    def dense_output(self, tq, smoothed: bool):
        """Posterior at the times `tq` for every trajectory: (mean [n_q, D, N], cov_tril [n_q, TRI, N])."""
        tq = np.ascontiguousarray(tq, dtype=np.float64)
        self._chk(self.lib.odef_dense_output(self._h, _as_dp(tq), len(tq), int(smoothed)))
        nb = self.field_bytes(F_DENSE_MEAN)
        m = np.empty(nb // 8)
        self._chk(self.lib.odef_get(self._h, F_DENSE_MEAN, m.ctypes.data_as(_vp), nb))
        nb = self.field_bytes(F_DENSE_COV_TRIL)
        c = np.empty(nb // 8)
        self._chk(self.lib.odef_get(self._h, F_DENSE_COV_TRIL, c.ctypes.data_as(_vp), nb))
        return m.reshape(len(tq), self.D, self.N), c.reshape(len(tq), self.TRI, self.N)

    def sample_states(self, n: int, seed: int, noise_scale: float = 1.0):
        """n joint posterior draws of the state path per trajectory: [n_save, D, n, N]."""
        self._chk(self.lib.odef_sample(self._h, int(n), int(seed) & 0xFFFFFFFFFFFFFFFF, float(noise_scale)))
        nb = self.field_bytes(F_SAMPLES)
        a = np.empty(nb // 8)
        self._chk(self.lib.odef_get(self._h, F_SAMPLES, a.ctypes.data_as(_vp), nb))
        return a.reshape(self.n_save, self.D, int(n), self.N)

    def dense_sample_states(self, tq, n: int, seed: int, noise_scale: float = 1.0):
        """n joint draws of the state path on the times tq (filter-interpolated states): [n_q, D, n, N]."""
        tq = np.ascontiguousarray(tq, float)
        self._chk(self.lib.odef_dense_sample(self._h, tq.ctypes.data_as(C.POINTER(C.c_double)), len(tq), int(n),
                                             int(seed) & 0xFFFFFFFFFFFFFFFF, float(noise_scale)))
        nb = self.field_bytes(F_SAMPLES)
        a = np.empty(nb // 8)
        self._chk(self.lib.odef_get(self._h, F_SAMPLES, a.ctypes.data_as(_vp), nb))
        return a.reshape(len(tq), self.D, int(n), self.N)

    def synchronize(self):
        self._chk(self.lib.odef_synchronize(self._h))

    @property
    def n_save(self) -> int:
        return int(self.lib.odef_n_save(self._h))

    def field_bytes(self, f: int) -> int:
        b = C.c_size_t()
        self._chk(self.lib.odef_field_bytes(self._h, f, C.byref(b)))
        return b.value

    def get(self, f: int) -> np.ndarray:
        """Field in the device layout (include/odefilter.h), as a flat numpy array reshaped."""
        nbytes = self.field_bytes(f)
        dt = np.int32 if f in _INT_FIELDS else np.float64
        out = np.empty(nbytes // np.dtype(dt).itemsize, dtype=dt)
        self._chk(self.lib.odef_get(self._h, f, out.ctypes.data_as(_vp), nbytes))
        ns, N = self.n_save, self.N
        if f in (F_MEAN, F_SMOOTH_MEAN):
            return out.reshape(ns, self.D, N)
        if f in (F_COV_TRIL, F_SMOOTH_COV_TRIL):
            return out.reshape(ns, self.TRI, N)
        if f == F_DIFFUSION:
            return out.reshape(ns, N)
        if f == F_T:
            return out.reshape(ns, N) if out.size == ns * N and out.size != ns else out
        if f == F_U0:
            return out.reshape(self.d, N)
        return out

    def device_ptr(self, f: int):
        p, b = _vp(), C.c_size_t()
        self._chk(self.lib.odef_get_device(self._h, f, C.byref(p), C.byref(b)))
        return p.value, b.value

    def bind_device(self, f: int, ptr: int, nbytes: int):
        self._chk(self.lib.odef_bind_device(self._h, f, _vp(ptr), nbytes))

    def kernel_name(self, which=0) -> str:
        """Name of the kernel the last filter (0) / smoother (1) pass launched, as a profiler prints it."""
        buf = C.create_string_buffer(256)
        self._chk(self.lib.odef_kernel_name(self._h, which, buf, 256))
        return buf.value.decode()

    def kernel_time_ms(self, which=0):
        ms, n = C.c_float(), C.c_int()
        self._chk(self.lib.odef_kernel_time_ms(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value


# ---- constants and step-level functions --------------------------------------------------


def ibm(d: int, q: int):
    """`ProbNumDiffEq.ibm(d, q)` (src/priors.jl:7-59): A (upper triangular), Q_L = chol(Q).L."""
    lib = load_library()
    D = d * (q + 1)
    A, QL = np.zeros((D, D)), np.zeros((D, D))
    if lib.odef_ibm(d, q, _as_dp(A), _as_dp(QL)) != 0:
        raise OdefError("odef_ibm failed")
    return A, QL


def preconditioner(d: int, q: int):
    """`ProbNumDiffEq.preconditioner(T, d, q)` (src/preconditioning.jl:1-17): returns P(h) -> diag."""
    lib = load_library()

    def P(h: float) -> np.ndarray:
        out = np.zeros(d * (q + 1))
        if lib.odef_preconditioner(d, q, float(h), _as_dp(out)) != 0:
            raise OdefError("odef_preconditioner failed")
        return out

    return P


def _batched(mu, L):
    mu = np.ascontiguousarray(np.atleast_2d(np.asarray(mu, float)))
    L = np.ascontiguousarray(np.asarray(L, float).reshape(mu.shape[0], mu.shape[1], mu.shape[1]))
    return mu, L


def predict(mu, L, A, Q_L):
    """`predict(x_curr, Ah, Qh)` (src/filtering.jl:56) on a batch; returns (mean, cov)."""
    lib = load_library()
    mu, L = _batched(mu, L)
    n, D = mu.shape
    A = np.ascontiguousarray(A, float); Q_L = np.ascontiguousarray(Q_L, float)
    mo, co = np.empty((n, D)), np.empty((n, D, D))
    if lib.odef_predict(D, n, _as_dp(mu), _as_dp(L), _as_dp(A), _as_dp(Q_L), _as_dp(mo), _as_dp(co)) != 0:
        raise OdefError("odef_predict failed")
    return mo, co


def update(mu_pred, L_pred, H, z):
    """`update(x_pred, measurement, H, R=0)` (src/filtering.jl:97) on a batch; returns (mean, cov)."""
    lib = load_library()
    mu, L = _batched(mu_pred, L_pred)
    n, D = mu.shape
    H = np.ascontiguousarray(np.asarray(H, float).reshape(n, -1, D))
    o = H.shape[1]
    z = np.ascontiguousarray(np.asarray(z, float).reshape(n, o))
    mo, co = np.empty((n, D)), np.empty((n, D, D))
    if lib.odef_update(D, o, n, _as_dp(mu), _as_dp(L), _as_dp(H), _as_dp(z), _as_dp(mo), _as_dp(co)) != 0:
        raise OdefError("odef_update failed")
    return mo, co


def smooth_step(mu, L, mu_s, L_s, A, Q_L):
    """`smooth(x_curr, x_next_smoothed, Ah, Qh)` (src/filtering.jl:136) on a batch; returns (mean, cov)."""
    lib = load_library()
    mu, L = _batched(mu, L)
    mu_s, L_s = _batched(mu_s, L_s)
    n, D = mu.shape
    A = np.ascontiguousarray(A, float); Q_L = np.ascontiguousarray(Q_L, float)
    mo, co = np.empty((n, D)), np.empty((n, D, D))
    if lib.odef_smooth_step(D, n, _as_dp(mu), _as_dp(L), _as_dp(mu_s), _as_dp(L_s), _as_dp(A), _as_dp(Q_L),
                            _as_dp(mo), _as_dp(co)) != 0:
        raise OdefError("odef_smooth_step failed")
    return mo, co


# ---- the solver interface (names of src/algorithms.jl and DiffEqBase) ---------------------


@dataclass(frozen=True)
class EK0:
    """`EK0(; prior=:ibm, order=3, diffusionmodel=:dynamic, smooth=true)` (src/algorithms.jl:23-28)."""
    prior: str = "ibm"
    order: int = 3
    diffusionmodel: str = "dynamic"
    smooth: bool = True
    _id = EK0_ID


@dataclass(frozen=True)
class EK1:
    """`EK1(; prior=:ibm, order=3, diffusionmodel=:dynamic, smooth=true)` (src/algorithms.jl:46-51)."""
    prior: str = "ibm"
    order: int = 3
    diffusionmodel: str = "dynamic"
    smooth: bool = True
    _id = EK1_ID


@dataclass
class ODEProblem:
    """`ODEProblem(f, u0, tspan, p)`.  `f` names a vector field of the device registry
    (a GPU kernel cannot call a host closure)."""
    f: str
    u0: Sequence[float]
    tspan: tuple
    p: Sequence[float] = ()

    def __post_init__(self):
        if np.ndim(self.u0) != 1:
            # src/caches.jl:46-49, test/errors.jl:11-14
            raise OdefError("Problems which are not scalar- or vector-valued (e.g. u0 is a scalar or a matrix) "
                            "are currently not supported")


@dataclass
class EnsembleProblem:
    """`EnsembleProblem(prob; prob_func)`: either explicit initial values `u0s[N, d]` (and
    optionally `ps[N, n_params]`), or the synthetic perturbation of SURVEY.md 8(d) generated
    on the device: u0_i = u0 + scale*(2U-1), U from splitmix64(seed + ...)."""
    prob: ODEProblem
    u0s: Optional[np.ndarray] = None
    ps: Optional[np.ndarray] = None
    perturb_scale: Optional[float] = None
    seed: int = 0x0DEF17E5
    first_index: int = 0
    n_perturbed: Optional[int] = None


@dataclass(frozen=True)
class EnsembleHIP:
    """Ensemble algorithm: all trajectories on one MI355X (`device`), one lane per trajectory.

    `distributed=True` (one process per GPU under torchrun): every rank solves the contiguous block of the trajectory
    index `dist.shard_bounds` assigns to it on the GPU `LOCAL_RANK`, nothing is exchanged while stepping, and
    `EnsembleSolution.gather_final()` is the single RCCL all-gather of the path."""
    device: int = -1
    distributed: bool = False
    backend: str = "nccl"  # torch.distributed backend of the distributed path ("nccl" = RCCL over xGMI; "gloo" for CPU tests)


def shard_ensemble(prob: "EnsembleProblem", trajectories: Optional[int], rank: int, world: int):
    """(per-rank EnsembleProblem, n_local, (lo, hi)): contiguous block of the trajectory index.  The synthetic ensemble
    keeps its GLOBAL numbering (first_index + lo), explicit u0s / ps are sliced -- so the union of the shards is exactly
    the ensemble one larger solve would produce."""
    from . import dist as od

    total = len(prob.u0s) if prob.u0s is not None else trajectories
    if total is None:
        raise OdefError("trajectories is required")
    lo, hi = od.shard_bounds(int(total), rank, world)
    if prob.u0s is not None:
        local = EnsembleProblem(prob.prob, u0s=np.asarray(prob.u0s)[lo:hi], ps=None if prob.ps is None else np.asarray(prob.ps)[lo:hi])
    else:
        local = EnsembleProblem(prob.prob, ps=None if prob.ps is None else np.asarray(prob.ps)[lo:hi],
                                perturb_scale=prob.perturb_scale, seed=prob.seed, first_index=prob.first_index + lo,
                                n_perturbed=prob.n_perturbed)
    return local, hi - lo, (lo, hi)


def fixed_time_grid(t0: float, t1: float, dt: float, tstops: Optional[Sequence[float]] = None) -> np.ndarray:
    """OrdinaryDiffEq's fixed-step grid: t += dt, every step clipped onto the next tstop (the end of the time span is
    one), with the 100-eps snap of the integrator loop."""
    if not (np.isfinite(dt) and dt > 0.0):
        raise OdefError(f"dt must be positive and finite, got {dt!r}")
    if not (t1 > t0):
        raise OdefError(f"tspan must be increasing, got ({t0!r}, {t1!r})")
    stops = sorted({float(x) for x in (tstops if tstops is not None else []) if t0 < float(x) < t1} | {float(t1)})
    ts = [t0]
    t = t0
    eps = np.finfo(float).eps
    for stop in stops:
        while t < stop:
            h = min(dt, stop - t)
            tn = t + h
            if abs(tn - stop) < 100 * eps * max(abs(tn), abs(stop)):
                tn = stop
            ts.append(tn)
            t = tn
    return np.array(ts)


def unpack_tril(c: np.ndarray, D: int) -> np.ndarray:
    """[..., TRI] -> [..., D, D] symmetric."""
    out = np.empty(c.shape[:-1] + (D, D))
    il = np.tril_indices(D)
    out[..., il[0], il[1]] = c
    out[..., il[1], il[0]] = c
    return out


@dataclass
class DEStats:
    nf: np.ndarray
    njacs: np.ndarray
    naccept: np.ndarray
    nreject: np.ndarray


class EnsembleSolution:
    """Per-trajectory `ProbODESolution` fields (src/solution.jl:8-24), batched.  Arrays are
    fetched from the device lazily and returned trajectory-major: mean[N, n_save, D]."""

    def __init__(self, ctx: Context, alg, adaptive: bool):
        self.ctx, self.alg, self.adaptive = ctx, alg, adaptive
        self.d, self.q, self.D = ctx.d, ctx.q, ctx.D
        self._cache = {}
        self.shard = None  # (lo, hi, total, world) of a distributed solve

    def final_mean(self) -> np.ndarray:
        """[N, D]: posterior mean of the state at each trajectory's last time (the filter state, which is also the
        smoothed one there, src/smoothing.jl:11)."""
        m = self._get(F_MEAN)  # [n_save, D, N]
        if not self.adaptive:
            return np.ascontiguousarray(m[-1].T)
        last = self._get(F_NSAVED).astype(np.int64) - 1
        return np.ascontiguousarray(m[last, :, np.arange(m.shape[2])])

    def gather_final(self) -> np.ndarray:
        """Distributed solves: the one collective of the path -- all-gather of the per-shard final means into
        [N_total, D] in global trajectory order on every rank (RCCL over xGMI; shards may differ by one row, so they
        are padded to the largest).  A non-distributed solve returns its own final means."""
        mine = self.final_mean()
        if self.shard is None or self.shard[3] == 1:
            return mine
        import torch

        from . import dist as od

        lo, hi, total, world = self.shard
        rows = -(-total // world)
        # the GPU this rank's context runs on (RCCL needs one distinct device per rank: never "the current device")
        dev = torch.device("cuda", int(self.ctx.cfg.device)) if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
        buf = torch.zeros((rows, self.D), dtype=torch.float64, device=dev)
        buf[: hi - lo] = torch.from_numpy(mine).to(dev)
        g = od.allgather_shards(buf, world).cpu().numpy()  # [world, rows, D]
        return np.concatenate([g[r, : od.shard_bounds(total, r, world)[1] - od.shard_bounds(total, r, world)[0]] for r in range(world)])

    def _get(self, f):
        if f not in self._cache:
            self._cache[f] = self.ctx.get(f)
        return self._cache[f]

    # Adaptive solves keep one device record per ATTEMPTED step; a rejected attempt repeats the previous record at
    # the unchanged time (include/odefilter.h, odef_solve_adaptive).  The reference saves accepted steps only
    # (src/integrator_utils.jl:33-48), so the accessors below drop the repeats.
    def _order(self):
        """(order [N, n_save], count [N]): per trajectory the record indices with the kept ones first."""
        if "_order" not in self._cache:
            tt = self._tsave()
            ns = self._get(F_NSAVED)
            idx = np.arange(tt.shape[1])[None, :]
            keep = idx < ns[:, None]
            keep[:, 1:] &= tt[:, 1:] != tt[:, :-1]
            self._cache["_order"] = (np.argsort(~keep, axis=1, kind="stable"), keep.sum(axis=1).astype(np.int32))
        return self._cache["_order"]

    def _tsave(self) -> np.ndarray:
        """Adaptive solves: the per-trajectory save times [N, n_save] (also for N = 1)."""
        return self._get(F_T).reshape(self.ctx.n_save, self.ctx.N).T

    def raw_index(self, i: int) -> np.ndarray:
        """Device record index of every kept record of trajectory i (identity for fixed grids)."""
        if not self.adaptive:
            return np.arange(self.ctx.n_save)
        order, count = self._order()
        return order[i, : count[i]]

    def _compact(self, a: np.ndarray) -> np.ndarray:
        """a: [N, n_save, ...] -> same shape, kept records first, the rest zero."""
        if not self.adaptive:
            return a
        order, count = self._order()
        out = np.take_along_axis(a, order.reshape(order.shape + (1,) * (a.ndim - 2)), axis=1)
        out[np.arange(a.shape[1])[None, :] >= count[:, None]] = 0
        return out

    @property
    def t(self) -> np.ndarray:
        """Fixed grid: [n_save] (shared by all trajectories); adaptive: [N, n_save], zero-padded."""
        return self._compact(self._tsave()) if self.adaptive else self._get(F_T).reshape(-1)

    @property
    def nsaved(self) -> np.ndarray:
        return self._order()[1] if self.adaptive else self._get(F_NSAVED)

    def x_filt_mean(self) -> np.ndarray:
        return self._compact(self._get(F_MEAN).transpose(2, 0, 1))

    def x_filt_cov(self) -> np.ndarray:
        return unpack_tril(self._compact(self._get(F_COV_TRIL).transpose(2, 0, 1)), self.D)

    def x_smooth_mean(self) -> np.ndarray:
        return self._compact(self._get(F_SMOOTH_MEAN).transpose(2, 0, 1))

    def x_smooth_cov(self) -> np.ndarray:
        return unpack_tril(self._compact(self._get(F_SMOOTH_COV_TRIL).transpose(2, 0, 1)), self.D)

    @property
    def smoothed(self) -> bool:
        return bool(self.alg.smooth) and self.ctx.cfg.save_mode == SAVE_EVERYSTEP

    @property
    def u(self) -> np.ndarray:
        """sol.u: posterior mean of the solution, smoothed when alg.smooth (src/integrator_utils.jl:20-26)."""
        m = self.x_smooth_mean() if self.smoothed else self.x_filt_mean()
        return m[:, :, : self.d]

    @property
    def pu_cov(self) -> np.ndarray:
        """Covariance of sol.pu = SolProj * x (src/integrator_utils.jl:21,45)."""
        c = self.x_smooth_cov() if self.smoothed else self.x_filt_cov()
        return c[:, :, : self.d, : self.d]

    def __call__(self, t, smoothed: Optional[bool] = None):
        """`sol(t)`: dense output (src/solution.jl:211-215).  Returns (mean [N, n_t, D], cov [N, n_t, D, D]) of the
        posterior at the times `t` (smoothed when the solution is, as the reference's interpolant)."""
        tq = np.atleast_1d(np.asarray(t, float))
        sm = self.smoothed if smoothed is None else smoothed
        m, c = self.ctx.dense_output(tq, sm)
        return m.transpose(2, 0, 1), unpack_tril(c.transpose(2, 0, 1), self.D)

    def sample_states(self, n: int = 1, seed: int = 0x5A3B1E, noise_scale: float = 1.0) -> np.ndarray:
        """`sample_states(sol, n)` (src/solution_sampling.jl:15-18, 24-62): [N, n_save, D, n] joint draws of the
        state path from the smoothing posterior.  The reference asserts a smoothing solve (:16)."""
        if not self.smoothed:
            raise AssertionError("sampling not implemented for non-smoothed posteriors")
        return self._compact(self.ctx.sample_states(n, seed, noise_scale).transpose(3, 0, 1, 2))

    def sample(self, n: int = 1, seed: int = 0x5A3B1E) -> np.ndarray:
        """`sample(sol, n)` (src/solution_sampling.jl:19-23): [N, n_save, d, n]."""
        return self.sample_states(n, seed)[:, :, : self.d, :]

    def dense_sample_states(self, n: int = 1, seed: int = 0x5A3B1E, times=None, noise_scale: float = 1.0):
        """`dense_sample_states(sol, n)` (src/solution_sampling.jl:63-69): ([N, n_t, D, n], times); the reference's
        grid is range(t0, t_end, length=1000), any non-decreasing `times` in [t0, t_end] may be given instead."""
        if not self.smoothed:
            raise AssertionError("sampling not implemented for non-smoothed posteriors")
        if times is None:
            t = self.t
            t0, t_end = (t[0, 0], t[0, self.nsaved[0] - 1]) if self.adaptive else (t[0], t[-1])
            times = np.linspace(float(t0), float(t_end), 1000)
        times = np.ascontiguousarray(times, float)
        return self.ctx.dense_sample_states(times, n, seed, noise_scale).transpose(3, 0, 1, 2), times

    def dense_sample(self, n: int = 1, seed: int = 0x5A3B1E, times=None):
        """`dense_sample(sol, n)` (src/solution_sampling.jl:70-75): ([N, n_t, d, n], times)."""
        s, times = self.dense_sample_states(n, seed, times)
        return s[:, :, : self.d, :], times

    @property
    def diffusions(self) -> np.ndarray:
        """[N, n_save-1]: entry k = diffusion of step t[k] -> t[k+1] (src/integrator_utils.jl:44)."""
        return self._compact(self._get(F_DIFFUSION).T)[:, 1:]

    @property
    def log_likelihood(self) -> np.ndarray:
        return self._get(F_LOGLIK)

    @property
    def destats(self) -> DEStats:
        return DEStats(self._get(F_NF), self._get(F_NJAC), self._get(F_NACCEPT), self._get(F_NREJECT))

    @property
    def retcode(self):
        return [RETCODES[int(r)] for r in self._get(F_RETCODE)]

    @property
    def retcode_raw(self) -> np.ndarray:
        return self._get(F_RETCODE)


def solve(prob, alg, ensemblealg: EnsembleHIP = EnsembleHIP(), *, trajectories: Optional[int] = None,
          dt: Optional[float] = None, adaptive: bool = True, abstol: float = 1e-6, reltol: float = 1e-3,
          tstops: Optional[Sequence[float]] = None, save_everystep: bool = True, maxiters: int = 100000,
          max_steps: Optional[int] = None, want_loglik: bool = True) -> EnsembleSolution:
    """`solve(EnsembleProblem(prob), EK1(order=3), EnsembleHIP(); trajectories, dt, adaptive, abstol, reltol)`.

    Keyword meaning follows DifferentialEquations.jl: `adaptive=false` needs `dt` (steps clipped onto `tstops` and the
    end of the time span) or `tstops` alone as the full grid; with `adaptive=true`, `dt` is the initial step.

    Differences from the reference's defaults, stated: (i) without `dt` an adaptive solve starts with 1e-3 (t1 - t0),
    not with OrdinaryDiffEq's automatic initial step (that heuristic evaluates `f` on the host; the vector field
    lives on the device) -- `sol.t` / `destats` then differ from the reference's, the posterior at common times does
    not beyond the tolerances; (ii) the device keeps one record per ATTEMPTED step, so the step budget is
    `max_steps` (default: `maxiters`, capped so that the records stay below 8 GiB); a trajectory that exhausts it ends
    with retcode MaxIters and a RuntimeWarning is raised, as for any other non-Success retcode."""
    if isinstance(prob, ODEProblem):
        prob = EnsembleProblem(prob, u0s=np.asarray(prob.u0, float)[None, :])
        trajectories = 1
    shard = None
    if ensemblealg.distributed:
        from . import dist as od

        rank, world, local_rank = od.init_from_env(backend=ensemblealg.backend)
        if ensemblealg.backend == "nccl":
            import torch

            torch.cuda.set_device(local_rank if ensemblealg.device < 0 else ensemblealg.device)
        total = len(prob.u0s) if prob.u0s is not None else trajectories
        prob, trajectories, (lo, hi) = shard_ensemble(prob, trajectories, rank, world)
        if hi == lo:
            raise OdefError(f"rank {rank} of {world} received no trajectory (ensemble of {total})")
        shard = (lo, hi, int(total), world)
        if ensemblealg.device < 0:
            ensemblealg = EnsembleHIP(device=local_rank, distributed=True, backend=ensemblealg.backend)
    base = prob.prob
    if alg.prior != "ibm":
        raise OdefError("Only the ibm prior is implemented so far")  # src/caches.jl:69
    if alg.diffusionmodel not in DIFFUSION:
        raise OdefError(f"diffusionmodel {alg.diffusionmodel!r} is not on the device path; use 'dynamic', 'fixed' or 'fixedMAP'")
    if not adaptive and dt is None and tstops is None:
        # test/errors.jl:17-19
        raise OdefError("Fixed timestep methods require a choice of dt or choosing the tstops")
    if prob.u0s is not None:
        u0s = np.ascontiguousarray(prob.u0s, float)
        N = u0s.shape[0]
        if trajectories is not None and trajectories != N:
            raise OdefError(f"trajectories={trajectories} but u0s has {N} rows")
    else:
        if trajectories is None:
            raise OdefError("trajectories is required")
        N = trajectories
    t0, t1 = float(base.tspan[0]), float(base.tspan[1])
    shared = prob.ps is None
    ctx = Context(base.f, alg.order, alg._id, N, diffusion=alg.diffusionmodel, smooth=alg.smooth,
                  save_everystep=save_everystep, params_shared=shared, device=ensemblealg.device,
                  want_loglik=want_loglik)
    p = np.asarray(base.p, float).reshape(-1) if shared else np.asarray(prob.ps, float)
    if prob.u0s is not None:
        ctx.set_problem(u0s, p, t0)
    else:
        if prob.perturb_scale is None:
            raise OdefError("EnsembleProblem needs u0s or perturb_scale")
        ctx.set_problem_perturbed(base.u0, p, t0, prob.perturb_scale, prob.seed, prob.first_index, prob.n_perturbed)
    if not (t1 > t0):
        raise OdefError(f"tspan must be increasing, got ({t0!r}, {t1!r})")
    if adaptive:
        if dt is not None and not (np.isfinite(dt) and dt > 0.0):
            raise OdefError(f"dt must be positive and finite, got {dt!r}")
        if max_steps is not None:
            ms = int(max_steps)
        else:  # maxiters attempts, bounded by the memory of one record per attempt
            rec_bytes = 8 * (ctx.D + ctx.TRI + 2) * (2 if alg.smooth else 1)
            ms = int(max(64, min(int(maxiters), (8 << 30) // (rec_bytes * N))))
        ctx.solve_adaptive(t1, abstol, reltol, dt if dt is not None else 1e-3 * (t1 - t0), None, ms)
    else:
        grid = np.asarray(tstops, float) if (tstops is not None and dt is None) else fixed_time_grid(t0, t1, dt, tstops)
        ctx.solve_fixed(grid)
    sol = EnsembleSolution(ctx, alg, adaptive)
    sol.shard = shard
    if alg.smooth and ctx.cfg.save_mode == SAVE_EVERYSTEP:
        ctx.smooth()
    bad = np.flatnonzero(sol.retcode_raw != 0)
    if bad.size:
        import warnings

        kinds = sorted({RETCODES[int(r)] for r in sol.retcode_raw[bad]})
        warnings.warn(f"{bad.size} of {N} trajectories did not finish with Success ({', '.join(kinds)}; first: trajectory "
                      f"{int(bad[0])}); see sol.retcode", RuntimeWarning, stacklevel=2)
    return sol
