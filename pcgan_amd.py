"""Import shim: `import pcgan_amd` loads the package in `promptable-counterfactual-gan_amd/` (a directory name that
is not a valid Python identifier) under the module name `pcgan_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "promptable-counterfactual-gan_amd")
_spec = importlib.util.spec_from_file_location("pcgan_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pcgan_amd"] = _mod
_spec.loader.exec_module(_mod)
