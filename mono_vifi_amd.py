"""Import shim: `import mono_vifi_amd` loads the package that lives in `mono-vifi_amd/`
(the directory name the project layout prescribes is not a valid Python identifier)."""
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_root, "mono-vifi_amd")
_spec = importlib.util.spec_from_file_location(
    "mono_vifi_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mono_vifi_amd"] = _mod
_spec.loader.exec_module(_mod)
