"""Import shim for the LAB-BENCH flavour: the same package source bound to ``lib/libsfm_amd_ab.so`` (``make ab``: the A/B
switches behind ``sfm_ransac_params.reserved[]``, the recorded slower kernel variants, the probe / trace hooks of
``include/sfm_amd_ab.h``).  Only tests/ and profiles/ import this; the product is ``import cuda_sfm_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cuda-sfm_amd")
_spec = importlib.util.spec_from_file_location(
    "cuda_sfm_amd_ab", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cuda_sfm_amd_ab"] = _mod
_spec.loader.exec_module(_mod)
