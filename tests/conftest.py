import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def gpu_ctx():
    """One svoh context on device 0.  Fails loudly (no skip, no fallback) when
    the HIP extension is missing or no GPU is present."""
    from svo_pro_universal_amd import frontend as fe
    ctx = fe.Context(0)
    yield ctx
    ctx.close()
