import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "emma-x_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# before any test makes the process's first HIP call (torch.cuda.is_available() in the `device` fixture): what `import emmax` sets
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


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


# ---- the id line ----------------------------------------------------------------------------------------------------
# "argmax == oracle wherever the margin exceeds twice the MEASURED error" cannot fail: it follows from the definition of the error.  The
# parity tests therefore draw the line A PRIORI: a greedy id must equal the fp32 oracle's wherever the oracle's top-2 margin exceeds twice a
# FIXED error budget (a fraction of max|logit| written down before the run), and the per-step error is asserted against its own tolerance
# separately.  An implementation whose error grows past the budget fails the id assertion on the first step whose margin sits between the two.
#   shallow (<= 3 decoder layers, bf16 operands): 1.5e-2 -- 3 x the 5e-3 these configurations measure (tiny configurations: 7.5e-3)
#   full depth (32 layers, bf16 operands):        4.5e-2 -- 1.3 x the worst step ever measured on this path (3.4e-2, rounds 3-5)
#   fp8 decode weights at full depth:             7.5e-2 -- 1.25 x its worst measured step (6.0e-2, round 4)
#   exact numerics (tuning switch exact):         1e-4   -- tests/test_exact_gpu.py, test_full_depth_gpu.py
ID_BUDGET_SHALLOW = 1.5e-2
ID_BUDGET_TINY = 7.5e-3        # the tiny configurations (hidden 256, 3 layers): 1.5 x the 5e-3 they measure (their random-weight margins are all below 2e-2)
ID_BUDGET_FULL_DEPTH = 4.5e-2
ID_BUDGET_FP8_FULL_DEPTH = 7.5e-2
ID_BUDGET_EXACT = 1e-4


def above_id_line(ref_row, budget):
    """True when the fp32 oracle's top-2 margin of this logit row exceeds 2 x budget x max|logit|: the id is then REQUIRED to match."""
    import torch

    top2 = torch.topk(ref_row.float(), 2).values
    return (top2[0] - top2[1]).item() > 2.0 * budget * ref_row.float().abs().max().item()
