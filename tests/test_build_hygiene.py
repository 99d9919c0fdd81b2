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


def _kernels(lines, substr):
    """(mangled name, body lines) of the kernels of an ISA listing whose name contains `substr`."""
    out, k = [], 0
    while k < len(lines):
        ln = lines[k]
        if ln.startswith("_Z") and substr in ln.split(":")[0] and ":" in ln:
            name = ln.split(":")[0]
            e = k + 1
            while e < len(lines) and not lines[e].startswith(".Lfunc_end"):
                e += 1
            out.append((name, lines[k:e]))
            k = e
        k += 1
    return out


def test_lane_smoother_owns_its_agpr_file():
    """csrc/smooth_lane.h keeps one packed matrix, the reciprocal pivots and the carried mean in a HAND-MANAGED file at the top
    of the lane's 256 AGPRs (inline assembly with fixed register numbers the compiler knows nothing about).  That is only
    sound while the compiler's own AGPR use -- it parks VGPRs there under pressure, lowest register first -- stays below the
    file: every AGPR reference outside the file's own assembly must lie below its first slot, and nothing may spill to scratch
    (a scratch spill of this kernel re-reads 100+ GB per pass)."""
    import re

    build = os.path.join(ROOT, "odefilters.jl_amd", "csrc", "build")
    if not glob.glob(os.path.join(build, "inst_smooth_d*.o")):
        pytest.skip("library objects not present")
    checked = 0
    for d in (2, 3):
        path = os.path.join(build, f"inst_smooth_d{d}-hip-amdgcn-amd-amdhsa-gfx950.s")
        assert os.path.exists(path), f"{path} missing: csrc/Makefile builds inst_smooth_d{d}.hip with --save-temps=obj"
        lines = open(path).read().split("\n")
        for name, body in _kernels(lines, "rts_smooth_lane_kernel"):
            m = re.search(r"rts_smooth_lane_kernelILi(\d+)ELi(\d+)ELb([01])E", name)
            dd, q = int(m.group(1)), int(m.group(2))
            D = dd * (q + 1)
            first_slot = 2 * (128 - (D * (D + 1) // 2 + D))  # MS of smooth_lane_v2, in 32-bit registers
            inasm, worst, own = False, -1, 0
            for ln in body:
                if "ASMSTART" in ln:
                    inasm = True
                elif "ASMEND" in ln:
                    inasm = False
                code = ln.split(";")[0]
                regs = [int(r) for r in re.findall(r"\ba(\d+)\b", code)]  # a12
                for grp in re.findall(r"\ba\[([0-9a-fx:]+)\]", code):  # a[12], a[0xdc] (the file's own accesses), a[4:5]
                    regs += [int(r, 0) for r in grp.split(":")]
                if not regs:
                    continue
                if inasm:
                    own += 1
                    assert min(regs) >= first_slot, f"{name}: the AGPR file reaches below its first slot: {ln.strip()}"
                else:
                    worst = max(worst, max(regs))
            assert own > 0, name
            assert worst < first_slot, f"{name}: the compiler uses a{worst}, the hand-managed file starts at a{first_slot}"
            checked += 1
        for ln in lines:
            m = re.match(r"\s*\.set (\S*rts_smooth_lane_kernel\S*)\.private_seg_size, (\d+)", ln)
            if m:
                assert int(m.group(2)) == 0, f"{m.group(1)} spills {m.group(2)} bytes per lane to scratch"
    assert checked >= 10
