"""Import shim: the package directory is named `gaussian-splatterer_amd` (not a Python identifier),
so load it by path and expose it as `gsplat_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gaussian-splatterer_amd")
_name = "gaussian_splatterer_amd"
if _name not in sys.modules:
    _spec = importlib.util.spec_from_file_location(_name, os.path.join(_dir, "__init__.py"),
                                                   submodule_search_locations=[_dir])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_name] = _mod
    _spec.loader.exec_module(_mod)
sys.modules[__name__] = sys.modules[_name]
