"""Import shim: the package directory is named ``osqp.jl_amd`` (after the
reference repo osqp/OSQP.jl), which is not a valid Python identifier.  This
module loads it under the importable name ``osqp_jl_amd``."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "osqp.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "osqp_jl_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["osqp_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
