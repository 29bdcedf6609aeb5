import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")
    config.addinivalue_line("markers", "grad: runs with torch autograd enabled (everything else runs under torch.no_grad())")


@pytest.fixture(autouse=True)
def _inference_tests_run_without_autograd(request):
    """The modules follow torch's own rule since round 3: gradients enabled + a parameter that requires one => a graph is recorded (the
    tokenizer's head, the projector, the splice) or the call is refused (inference-only modules).  Parity tests of the inference path
    therefore say what they mean and run under no_grad, as the reference's own inference callers do; tests marked `grad` exercise the
    differentiable boundary."""
    import torch
    if request.node.get_closest_marker("gpu") is None:                 # CPU tests never touch the HIP modules: left alone
        yield
    elif request.node.get_closest_marker("grad") is not None:
        with torch.enable_grad():
            yield
    else:
        with torch.no_grad():
            yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
