import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "structure-from-motion_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


@pytest.fixture(params=["early loops", "every pair through the full tests"])
def mixed_form(request, monkeypatch):
    """Runs a test once per form of the exact matrix-core scorer (ops.eh_score_mixed): with its early loops (pairs that fail a
    cheaper implied test are counted out without the full test) and with every pair through the full float32 tests."""
    from sfm_hip import ops

    monkeypatch.setattr(ops, "EVERY_PAIR_FULL_TEST", request.param != "early loops")
    return request.param


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
