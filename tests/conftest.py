import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_sessionstart(session):
    """The C-ABI library is a build artefact (git-ignored).  On a checkout where __graft_entry__.build() has not run yet, build it
    once (hipcc cross-compiles gfx950 without a GPU); if that is impossible the tests that need it fail loudly on their own."""
    lib = os.path.join(ROOT, "speechclip_plus_amd", "csrc", "libspeechclip_hip.so")
    if not os.path.exists(lib):
        try:
            from speechclip_plus_amd.build import build
            build(verbose=False)
        except Exception as e:       # noqa: BLE001 - reported, not hidden: the symbol-export test will fail with the real cause
            print(f"[conftest] could not build libspeechclip_hip.so: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))

    return load


def weights_from(fx, prefix="W_"):
    import torch

    return {k[len(prefix):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(prefix)}
