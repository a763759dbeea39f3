"""Import shim: the package directory is `krylov.jl_amd/` (a dot is not importable), so
`import krylov_jl_amd` loads it from there."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "krylov.jl_amd")
_spec = importlib.util.spec_from_file_location("krylov_jl_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["krylov_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
