import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hr():
    """The product library through its Python mirror.  No fallback: missing library => error."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from hybrid_rendering_amd import api
    api.lib()
    return api


@pytest.fixture(scope="session")
def ctx(hr):
    return hr.Context(0)
