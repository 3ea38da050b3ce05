import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "refnn_live: runs upstream's NN runtime (oracle/_ref/xqref_nn) on the GPU; opt-in, XQ_RUN_REFERENCE_NN=1")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
