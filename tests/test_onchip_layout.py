"""Index arithmetic of the on-chip smoother kernels (csrc/smooth_onchip.h, csrc/smooth_mfma.h), checked on the CPU: the
ownership of the result tiles of R = G M G' (every unordered pair of tile columns exactly once), the LDS swizzle (a
permutation of the tile under which both fragment reads are bank-conflict free), the tile-major hand-over layout of B, Y'
and the carried Sigma^s, and the LDS budget -- for every tile count the instantiated shapes and run-time compiled fields
can produce.  The functions are `__host__ __device__ constexpr`: hipcc compiles the host side only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_onchip_smoother_index_arithmetic(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = tmp_path / "onchip_layout_check"
    src = os.path.join(ROOT, "tests", "host", "onchip_layout_check.hip")
    inc = os.path.join(ROOT, "odefilters.jl_amd", "csrc")
    r = subprocess.run([hipcc, "--cuda-host-only", "-O1", "-std=c++20", "-I", inc, src, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
