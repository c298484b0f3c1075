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


@pytest.fixture(scope="session")
def small():
    return dict(np.load(os.path.join(GOLDEN, "small.npz")))


@pytest.fixture(scope="session")
def full_summary():
    import json
    with open(os.path.join(GOLDEN, "full_summary.json")) as f:
        return json.load(f)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
