"""Import shim: `import hypergen_amd` loads the package in ./hyper-gen_amd/ (the directory keeps
the reference's crate name, which is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hyper-gen_amd")
_spec = importlib.util.spec_from_file_location(
    "hypergen_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hypergen_amd"] = _mod
_spec.loader.exec_module(_mod)
