"""Import shim: the package directory is `neuralrecon-w_amd/` (hyphenated, as the project layout
prescribes), which Python cannot import by name.  `import neuralrecon_w_amd` loads it."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "neuralrecon-w_amd")
_spec = importlib.util.spec_from_file_location(
    "neuralrecon_w_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["neuralrecon_w_amd"] = _mod
_spec.loader.exec_module(_mod)
