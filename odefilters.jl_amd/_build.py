"""Build libodefilter_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libodefilter_hip.so")


def build(jobs: int = 8, verbose: bool = False) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0 or verbose:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libodefilter_hip.so failed")
    return LIB


def is_built() -> bool:
    return os.path.exists(LIB)
