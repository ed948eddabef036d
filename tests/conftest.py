import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # tests/_margins.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_sessionfinish(session, exitstatus):
    """Measured-vs-bound table of the tolerance checks that go through tests/_margins.within."""
    try:
        from _margins import MARGINS
    except Exception:
        return
    if not MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "margins.json"), "w") as f:
        json.dump(MARGINS, f, indent=1, sort_keys=True)


@pytest.fixture(autouse=True)
def _device_quiesced_between_tests(request):
    """A GPU test leaves no work in flight: a device fault is then reported against the test that launched the faulting
    kernel, not against whichever later test first synchronises."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
