import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def lrp():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("image-lens-reproject_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding

    return oracle_binding


@pytest.fixture(scope="session")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("a -m gpu test ran without a GPU: the HIP path has no fallback")
    return torch
