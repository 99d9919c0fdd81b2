"""Build hygiene of libodefilter_hip.so's device code (CPU suite: needs only the ROCm binutils and the built objects).

Every device function must be inlined into its kernels.  A function the inliner leaves out of line is compiled ONCE, for
the loosest register budget among its callers; called from a kernel with a tighter `__launch_bounds__` occupancy (the
D = 168 smoother runs four workgroups per CU = 128 registers, dense output and sampling two = 256) it addresses registers
the wavefront was never allocated: a memory access fault at address 0 on the GPU, invisible at compile time.  This
happened in round 2 (csrc/mfma_dense.h header); the functions are force-inlined since, and this test keeps it so."""
import glob
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _device_functions(obj):
    """(name, size) of the FUNC symbols in the gfx950 code object bundled into host object `obj`."""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [f for f in glob.glob(local + ".*") if "amdgcn" in f]
        if not cos:  # a host-only translation unit (jit.hip)
            return []
        out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", cos[0]], check=True, capture_output=True, text=True).stdout
    funcs = []
    for line in out.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            funcs.append((f[7], int(f[2])))
    return funcs


def test_every_device_function_is_inlined_into_its_kernels():
    objs = sorted(o for o in glob.glob(os.path.join(ROOT, "odefilters.jl_amd", "csrc", "build", "*.o"))
                  if "-hip-amdgcn" not in o and "-host-" not in o)  # (not the temporaries of a -save-temps build)
    if not objs or not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("library objects or ROCm binutils not present")
    stray = {}
    n_kernels = 0
    for o in objs:
        for name, size in _device_functions(o):
            if "kernel" in name:
                n_kernels += 1
            else:
                stray.setdefault(os.path.basename(o), []).append(name)
    assert n_kernels > 50
    assert not stray, f"device functions left out of line (force-inline them): {stray}"
