import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """(torch, device, Context) on cuda:0; GPU tests must never silently fall back to the CPU."""
    import torch
    assert torch.cuda.is_available(), "GPU test running without a GPU"
    import cuda_sfm_amd as S
    ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    return torch, torch.device("cuda:0"), ctx


@pytest.fixture(scope="session")
def gpu_ab():
    """The same for the lab-bench flavour of the library (libsfm_amd_ab.so, tests/test_gpu_ab.py)."""
    import torch
    assert torch.cuda.is_available(), "GPU test running without a GPU"
    import cuda_sfm_amd_ab as A
    ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
    return torch, torch.device("cuda:0"), ctx
