import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lean: runs the library's default (lean) arithmetic")


@pytest.fixture(autouse=True)
def _arithmetic(request, monkeypatch):
    """The gray IMC kernels come in two arithmetic variants (include/jaybenne_amd.h): the bit-parity
    tests run the exact one, whose results equal the oracle's bit for bit; tests marked ``lean``
    run the library's default and state its tolerance."""
    if request.node.get_closest_marker("lean"):
        monkeypatch.delenv("JB_EXACT_ARITH", raising=False)
    else:
        monkeypatch.setenv("JB_EXACT_ARITH", "1")


def pytest_collection_modifyitems(config, items):
    # a GPU test that stops making progress (a kernel that never drains) must end the run, not
    # sit on the box until the pool's own watchdog fires
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(300, method="thread"))


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no GPU is visible")
    return torch.device("cuda", 0)
