import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "emma-x_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("`-m gpu` tests need a HIP device; none is visible")
    return "cuda:0"


@pytest.fixture
def tune():
    """`tune(graph=1, ks=0)`: set tuning switches of libemmax_hip.so through its setter (the library reads the environment only once,
    at start-up); the previous values come back after the test."""
    from emmax import _lib

    undo = []

    def set_(**kw):
        for k, v in kw.items():
            undo.append((k, _lib.tuning_get(k)))
            _lib.tuning_set(k, v)

    yield set_
    for k, v in reversed(undo):
        _lib.tuning_set(k, v)
