import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import odefilters_jl_amd

    return odefilters_jl_amd


@pytest.fixture(scope="session")
def orc():
    import odefilter_oracle

    return odefilter_oracle


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first(request):
    """GPU runs only: bring up torch's (bundled) HIP runtime BEFORE libodefilter_hip.so creates its context.  Two HIP
    runtimes share the process in the bind/stream test; initialising torch second has been seen to fail with
    "No HIP GPUs are available" on a fresh box."""
    expr = request.config.getoption("-m") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # the tests that need torch report it themselves
            pass
