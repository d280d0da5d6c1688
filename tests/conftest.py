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
    """torch first.  libgs_hip.so links the HIP runtime by SONAME (libamdhip64.so.7) and torch bundles a copy under the
    same SONAME: with torch imported first -- what bench.py does -- the library binds torch's copy and the
    process holds ONE runtime (and, for RCCL, torch's librccl); with the library first the process holds /opt/rocm's
    runtime and torch's own copy then finds no GPU ("No HIP GPUs are available": tests/test_gpu_multiprocess.py pins all
    three orders in fresh interpreters).  The tests that compare planes on the device through torch views need torch's
    CUDA side, so torch's runtime comes up here, once per session, whichever test runs first.  No-op without a GPU.
    GS_TEST_TORCH_FIRST=0 switches it off (diagnostics)."""
    if os.environ.get("GS_TEST_TORCH_FIRST", "1") != "0":
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    yield
