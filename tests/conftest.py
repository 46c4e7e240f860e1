import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` under gpurun)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Make sure the native library and the oracle exist before any test runs (build is a no-op
    when up to date; on the GPU box the prebuilt .so files travel with the snapshot)."""
    from bevfusion_amd import _capi

    if not os.path.exists(_capi.LIB_PATH):
        from bevfusion_amd import build as _b

        _b.build()
    import oracle

    oracle.build()
    yield
