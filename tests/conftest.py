import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The native library is git-ignored: build it in-tree when a fresh checkout lacks it (hipcc
    # cross-compiles gfx950 without a GPU; on the GPU box the prebuilt .so travels with the repo).
    from pecanpy_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

