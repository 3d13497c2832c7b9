import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "engine_path_auto: leave the engine's kernel choice (fused / batch-level) at its default")


@pytest.fixture(scope="session")
def golden():
    return {ds: np.load(os.path.join(GOLDEN, f"{ds}_golden.npz")) for ds in ("ted", "beat")}


def max_abs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


@pytest.fixture(autouse=True)
def _fused_kernel_by_default(request):
    """The parity suites written for the fused step kernel (one workgroup per sample) use tiny batches, which the engine would now
    run on the batch-level kernels (`auto`): pin them to the fused kernel.  tests/test_gpu_smallbatch.py covers `batch` and `auto`
    against the same fixtures and opts out with the `engine_path_auto` marker."""
    from livelyspeaker_amd import _lib
    if request.node.get_closest_marker("engine_path_auto"):
        yield
        return
    old = _lib.Engine.default_path
    _lib.Engine.default_path = "fused"
    try:
        yield
    finally:
        _lib.Engine.default_path = old
