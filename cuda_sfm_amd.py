"""Import shim: the package directory is named ``cuda-sfm_amd`` (hyphen, as the project layout
prescribes), which is not a valid Python identifier.  ``import cuda_sfm_amd`` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cuda-sfm_amd")
_spec = importlib.util.spec_from_file_location(
    "cuda_sfm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cuda_sfm_amd"] = _mod
_spec.loader.exec_module(_mod)
