import os
import sys

import pytest

try:   # before libsvo_hip is loaded anywhere in the session: one HIP runtime for both (_capi._share_hip_runtime_with_torch),
    import torch  # noqa: F401  -- and never a first `import torch` in the middle of a GPU test
except ImportError:
    pass

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
    _live_ctx.append(ctx)
    yield ctx
    _live_ctx.clear()
    ctx.close()


_live_ctx = []


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment():
    """The library reads its SVOH_* tuning knobs when a context is made, never in a launch path
    (svoh_reload_knobs, include/svo_hip.h).  Tests that force a kernel geometry change the environment and call
    ctx.reload_knobs(); after a test -- and after monkeypatch has put the environment back -- the session's context
    reads them again, so no test inherits another's geometry."""
    yield
    for ctx in _live_ctx:
        ctx.reload_knobs()
