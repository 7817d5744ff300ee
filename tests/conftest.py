import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vit-lens_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (build container only)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    gpu = _has_gpu()
    have_ref = os.path.isdir("/root/reference/vitlens/src/open_clip")
    for it in items:
        if "gpu" in it.keywords and not gpu:
            it.add_marker(pytest.mark.skip(reason="no GPU in this container"))
        if "needs_reference" in it.keywords and not have_ref:
            it.add_marker(pytest.mark.skip(reason="/root/reference not present"))
