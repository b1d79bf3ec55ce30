"""Import shim: the package directory is named `tfhe.jl_amd/` (not importable by that literal name
because of the dot).  `import tfhe_jl_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tfhe.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "tfhe_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tfhe_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
