"""Import shim: the package directory name (fixed by the build contract) contains hyphens, so it
is loaded here by path and registered as the importable module `edrl_amd`."""
import importlib.util
import os
import sys

_NAME = "robust-multimodal-learning-for-ophthalmic-disease-grading-via-disentangled-representation_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), _NAME)

if "edrl_amd_pkg" not in sys.modules:
    _spec = importlib.util.spec_from_file_location("edrl_amd_pkg", os.path.join(_DIR, "__init__.py"),
                                                   submodule_search_locations=[_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["edrl_amd_pkg"] = _mod
    _spec.loader.exec_module(_mod)

_pkg = sys.modules["edrl_amd_pkg"]
sys.modules[__name__] = _pkg
