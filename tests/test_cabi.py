"""C-ABI boundary checks that need no GPU: the library loads, exports every symbol that
include/odefilter.h declares, host-side constant generators match the oracle, and the
product path fails loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "odefilter.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(odef_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_all_exported(pkg):
    lib = pkg.load_library()
    names = declared_symbols()
    assert len(names) == 44  # 28 per-context entry points + 16 of the multi-GPU group (odef_shard_range, odef_group_*, odef_allgather)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/odefilter.h but not exported"
    # and the Python binding table covers exactly the header
    from odefilters_jl_amd import host

    assert sorted(host.SYMBOLS) == names


def test_version_and_struct_size(pkg):
    from odefilters_jl_amd import host

    assert pkg.load_library().odef_version() == 100
    assert C.sizeof(host.OdefConfig) == 56 and C.sizeof(host.OdefController) == 80


@pytest.mark.parametrize("d,q", [(1, 2), (2, 3), (3, 3), (2, 5)])
def test_ibm_constants_match_oracle(pkg, orc, d, q):
    """src/priors.jl:7-59 through odef_ibm; KAT of test/priors.jl:50-59 for (1,2)."""
    A, QL = pkg.ibm(d, q)
    Ao, QLo = orc.ibm(d, q)
    np.testing.assert_array_equal(A, Ao)
    np.testing.assert_allclose(QL, QLo, rtol=0, atol=5e-12)  # Q is Hilbert-like (cond 1.7e10 at q=5): LAPACK and this chol are equally far (3e-12) from the exact factor
    if (d, q) == (1, 2):
        np.testing.assert_allclose(QL @ QL.T, [[1 / 20, 1 / 8, 1 / 6], [1 / 8, 1 / 3, 1 / 2], [1 / 6, 1 / 2, 1]], rtol=1e-14)


def test_preconditioner_matches_oracle(pkg, orc):
    """src/preconditioning.jl:1-17 through odef_preconditioner (bit-identical running product)."""
    for h in (2.0**-9, 7e-2, 0.05, 1e-4):
        np.testing.assert_array_equal(pkg.preconditioner(3, 3)(h), orc.preconditioner(3, 3)(h))


def test_no_cpu_fallback(pkg):
    """Without a GPU the product path must raise, not compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.OdefError, match="no HIP device"):
        pkg.Context("lorenz63", 3, 1, 64)
    with pytest.raises(pkg.OdefError):
        pkg.solve(pkg.ODEProblem("fhn", [-1.0, 1.0], (0.0, 1.0), (0.2, 0.2, 3.0)), pkg.EK0(order=1), dt=0.1, adaptive=False)


def test_create_argument_validation(pkg):
    from odefilters_jl_amd import host

    lib = pkg.load_library()
    cfg = host.OdefConfig()
    h = C.c_void_p()
    assert lib.odef_create(C.byref(h), C.byref(cfg)) != 0  # struct_size = 0
    assert b"struct_size" in lib.odef_last_error(None)
    cfg.struct_size = C.sizeof(host.OdefConfig)
    cfg.rhs_id, cfg.d, cfg.n_params, cfg.order, cfg.n_traj = 1, 2, 3, 3, 8  # wrong d for Lorenz
    assert lib.odef_create(C.byref(h), C.byref(cfg)) != 0
    assert b"dimension" in lib.odef_last_error(None)
    cfg.d, cfg.order = 3, 9
    assert lib.odef_create(C.byref(h), C.byref(cfg)) != 0
    assert b"order" in lib.odef_last_error(None)


def test_shard_range_arithmetic(pkg):
    """odef_shard_range (host arithmetic of the multi-GPU group, no GPU needed): contiguous, ordered, complete, the
    first n % G shards one longer -- the partition SURVEY.md 8(e) prescribes; dist.shard_bounds is the same function."""
    from odefilters_jl_amd import dist, host

    for n, G in [(65536, 8), (16384, 8), (7, 2), (10, 3), (8, 8), (1000003, 6)]:
        nxt = 0
        for g in range(G):
            first, cnt = host.shard_range(n, G, g)
            assert first == nxt and cnt in (n // G, n // G + 1) and (cnt == n // G + 1) == (g < n % G)
            assert dist.shard_bounds(n, g, G) == (first, first + cnt)
            nxt = first + cnt
        assert nxt == n
    with pytest.raises(host.OdefError):
        host.shard_range(10, 3, 3)
    with pytest.raises(host.OdefError):
        host.shard_range(10, 0, 0)


def test_struct_layout_from_a_c_translation_unit(pkg, tmp_path):
    """A C (not ctypes) program compiled against include/odefilter.h prints sizeof / offsetof of odef_config and
    odef_controller and the values of the field / return-code enums; the ctypes mirror (host.py) and the Julia struct
    (julia/ODEFilterHIP.jl, parsed as text: no Julia in the image) must describe the same layout."""
    import re
    import subprocess

    from odefilters_jl_amd import host

    cfg_fields = [f for f, _ in host.OdefConfig._fields_]
    ctl_fields = [f for f, _ in host.OdefController._fields_]
    src = ['#include <stddef.h>', '#include <stdio.h>', '#include "odefilter.h"', "int main(void) {",
           '  printf("sizeof odef_config %zu\\n", sizeof(odef_config));', '  printf("sizeof odef_controller %zu\\n", sizeof(odef_controller));']
    src += [f'  printf("odef_config.{f} %zu %zu\\n", offsetof(odef_config, {f}), sizeof(((odef_config*)0)->{f}));' for f in cfg_fields]
    src += [f'  printf("odef_controller.{f} %zu %zu\\n", offsetof(odef_controller, {f}), sizeof(((odef_controller*)0)->{f}));' for f in ctl_fields]
    enums = ["ODEF_F_MEAN", "ODEF_F_COV_TRIL", "ODEF_F_DIFFUSION", "ODEF_F_T", "ODEF_F_LOGLIK", "ODEF_F_NACCEPT", "ODEF_F_NREJECT", "ODEF_F_NF",
             "ODEF_F_NJAC", "ODEF_F_NSAVED", "ODEF_F_RETCODE", "ODEF_F_SMOOTH_MEAN", "ODEF_F_SMOOTH_COV_TRIL", "ODEF_F_U0", "ODEF_F_SAMPLES",
             "ODEF_RET_SUCCESS", "ODEF_RET_MAXITERS", "ODEF_RET_DT_LESS_THAN_MIN", "ODEF_RET_UNSTABLE", "ODEF_EK0", "ODEF_EK1", "ODEF_MAX_ORDER"]
    src += [f'  printf("{e} %d\\n", (int){e});' for e in enums] + ["  return 0;", "}"]
    cfile, exe = tmp_path / "layout.c", tmp_path / "layout"
    cfile.write_text("\n".join(src))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    out = {ln.split()[0]: ln.split()[1:] for ln in lines if not ln.startswith("sizeof")}
    sizes = dict(ln.split()[1:] for ln in lines if ln.startswith("sizeof"))
    assert int(sizes["odef_config"]) == C.sizeof(host.OdefConfig) and int(sizes["odef_controller"]) == C.sizeof(host.OdefController)
    for cls, name in ((host.OdefConfig, "odef_config"), (host.OdefController, "odef_controller")):
        for f, t in cls._fields_:
            off, size = (int(v) for v in out[f"{name}.{f}"])
            assert off == getattr(cls, f).offset and size == C.sizeof(t), (name, f)
    # the Python constants
    for k, e in enumerate(enums[:13]):
        assert int(out[e][0]) == k
    assert int(out["ODEF_F_U0"][0]) == host.F_U0 and int(out["ODEF_F_SAMPLES"][0]) == host.F_SAMPLES
    assert [host.RETCODES[int(out[e][0])] for e in enums[15:19]] == ["Success", "MaxIters", "DtLessThanMin", "Unstable"]
    assert (int(out["ODEF_EK0"][0]), int(out["ODEF_EK1"][0])) == (host.EK0_ID, host.EK1_ID) and int(out["ODEF_MAX_ORDER"][0]) == host.MAX_ORDER
    # the Julia struct: same fields, same order, Int32 / Int64 where the C struct has 4 / 8 bytes
    jl = open(os.path.join(ROOT, "julia", "ODEFilterHIP.jl")).read()
    body = re.search(r"struct OdefConfig[^\n]*\n(.*?)\nend", jl, re.S).group(1)
    jfields = re.findall(r"(\w+)::(Int32|Int64)", body)
    assert [f for f, _ in jfields] == cfg_fields
    for f, t in jfields:
        assert int(out[f"odef_config.{f}"][1]) == {"Int32": 4, "Int64": 8}[t], f
    assert re.search(r"F_SAMPLES = (\d+)", jl).group(1) == out["ODEF_F_SAMPLES"][0]


@pytest.mark.parametrize("N,G", [(65536, 8), (65537, 8), (65543, 8), (1001, 8), (8, 8), (13, 3), (5, 1)])
def test_group_layout_and_unpadding_without_a_device(pkg, N, G):
    """The arithmetic of the multi-GPU group that has never met more than one device (SCALE runs were skipped in every round):
    shard ranges, the longest shard, the zero-padded equal blocks RCCL's all-gather moves, and the un-padding of the gathered
    block -- through the C ABI, for eight devices and ensemble sizes that do not divide evenly.  A synthetic gathered buffer
    (value = global trajectory number and state row, NaN in the padding) must come back as [D][N] with no NaN and every
    trajectory in its place."""
    lib = pkg.load_library()
    D = 12
    first, count = (C.c_int64 * G)(), (C.c_int64 * G)()
    cnt_max, blk = C.c_int64(), C.c_int64()
    assert lib.odef_group_layout(N, G, D, first, count, C.byref(cnt_max), C.byref(blk)) == 0
    first, count = list(first), list(count)
    assert sum(count) == N and first[0] == 0 and all(first[k + 1] == first[k] + count[k] for k in range(G - 1))
    assert max(count) - min(count) <= 1 and sorted(count, reverse=True) == count  # the first N % G shards are one longer
    assert cnt_max.value == max(count) == -(-N // G) and blk.value == D * cnt_max.value
    for k in range(G):  # the same ranges as odef_shard_range and as the torch.distributed host (dist.shard_bounds)
        f, c = C.c_int64(), C.c_int64()
        assert lib.odef_shard_range(N, G, k, C.byref(f), C.byref(c)) == 0 and (f.value, c.value) == (first[k], count[k])
    gathered = np.full((G, D, cnt_max.value), np.nan)
    for k in range(G):
        gathered[k, :, : count[k]] = (first[k] + np.arange(count[k]))[None, :] + 1e6 * np.arange(D)[:, None]
    dst = np.full((D, N), -1.0)
    dp = C.POINTER(C.c_double)
    assert lib.odef_unpad_gathered(gathered.ctypes.data_as(dp), G, D, N, dst.ctypes.data_as(dp)) == 0
    np.testing.assert_array_equal(dst, np.arange(N)[None, :] + 1e6 * np.arange(D)[:, None])


def test_group_layout_argument_validation(pkg):
    lib = pkg.load_library()
    z = C.c_int64()
    assert lib.odef_group_layout(7, 8, 12, None, None, C.byref(z), C.byref(z)) != 0  # fewer trajectories than devices
    assert lib.odef_group_layout(8, 0, 12, None, None, C.byref(z), C.byref(z)) != 0
    assert lib.odef_group_layout(8, 2, 0, None, None, C.byref(z), C.byref(z)) != 0
    assert lib.odef_group_layout(9, 2, 3, None, None, None, None) == 0  # every output is optional
    dp = C.POINTER(C.c_double)
    assert lib.odef_unpad_gathered(None, 2, 3, 9, np.zeros(27).ctypes.data_as(dp)) != 0
    # without a HIP device a group cannot be created: a clear error, not a crash
    cfg = pkg.host.OdefConfig(struct_size=C.sizeof(pkg.host.OdefConfig), alg=1, order=3, rhs_id=1, d=3, n_params=3, params_shared=1,
                              save_mode=0, device=-1, want_loglik=0, n_traj=64)
    import torch

    if not torch.cuda.is_available():
        h = C.c_void_p()
        assert lib.odef_group_create(C.byref(h), C.byref(cfg), 8, None) != 0
        assert b"device" in lib.odef_group_last_error(None).lower() or b"hip" in lib.odef_group_last_error(None).lower()
