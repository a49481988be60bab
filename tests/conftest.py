import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))

    return load


def weights_from(fx, prefix="W_"):
    import torch

    return {k[len(prefix):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(prefix)}
