import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of GPU time (config 5 solved to optimality); skipped unless ELLP_SLOW=1")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if os.environ.get("ELLP_SLOW") != "1":
        slow = pytest.mark.skip(reason="slow: minutes of GPU time; run with ELLP_SLOW=1 (result of the last run: profiles/)")
        for item in items:
            if "slow" in item.keywords:
                item.add_marker(slow)
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
