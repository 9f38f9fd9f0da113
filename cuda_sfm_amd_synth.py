"""Loads cuda-sfm_amd/synth.py WITHOUT importing the package (which requires the built HIP
library and torch): the synthetic-input generator is pure numpy and is also needed by CPU-only
tooling (tests/gen_golden.py, oracle tests)."""
import importlib.util
import os

_p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cuda-sfm_amd", "synth.py")
_spec = importlib.util.spec_from_file_location("cuda_sfm_amd_synth_impl", _p)
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)
