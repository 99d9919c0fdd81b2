"""Import shim: the package directory is named `odefilters.jl_amd` (not a valid Python
identifier), so it is registered under the importable name `odefilters_jl_amd`."""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(_HERE, "odefilters.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "odefilters_jl_amd", os.path.join(_PKG, "__init__.py"), submodule_search_locations=[_PKG]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["odefilters_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
