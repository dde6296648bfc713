import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU: skip instead of failing inside HIP.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd missing)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
