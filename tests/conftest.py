import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build libgs_hip.so and the oracle libraries once per session (hipcc cross-compiles)."""
    import __graft_entry__

    __graft_entry__.build()
    return True


@pytest.fixture(scope="session", autouse=True)
def torch_runtime_first():
    """A GPU test process holds TWO HIP runtimes: libgs_hip.so links /opt/rocm's, torch brings its own copy.
    bench.py and smoke() always bring torch's up first (torch.cuda.is_available / set_device before the first
    gs_ctx_create); the tests that compare planes on the device through torch views do the same here, once per
    session, so that the order never depends on which test happens to run first.  No-op without a GPU.
    GS_TEST_TORCH_FIRST=0 switches it off (diagnostics)."""
    if os.environ.get("GS_TEST_TORCH_FIRST", "1") != "0":
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    yield
