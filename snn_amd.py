"""Import shim: the package directory is `spiking-neural-networks_amd/` (hyphenated, not a valid
module name), so it is loaded by path and published as `snn_amd`."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spiking-neural-networks_amd")
_spec = importlib.util.spec_from_file_location(
    "snn_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["snn_amd"] = _mod
_spec.loader.exec_module(_mod)
