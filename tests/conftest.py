import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    """Build the native pieces once per session if they are missing or stale (hipcc and gcc are in
    the image on both the build container and the GPU box).  The product itself never builds
    implicitly: a missing library is a loud error there."""
    try:
        from warp_rnnt_amd import _build
        _build.build()
        _build.build_binding()
        import oracle
        oracle.build()
    except Exception as e:   # let the individual tests report the problem
        print("native build failed in conftest:", e)
