import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# seeded random weights stand in for checkpoints in tests (the product default is to refuse: see models/llama.py)
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")


def pytest_collection_modifyitems(config, items):
    """On a box without a HIP device `pytest tests` must show the CPU results, not hundreds of CUDA errors: gpu-marked
    tests are skipped there (the driver runs them with -m gpu on an MI355X)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X): run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
