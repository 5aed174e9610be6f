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


# ---- parity facts: numbers a test MEASURED (exact arg-max agreements, near-tie positions, accept lengths) are printed in
# the terminal summary -- i.e. in the tail of `pytest -q` that the driver keeps -- and written to gpurun_out/parity_facts.json,
# so a run that passes with 21 of 24 exact tokens can be told from one that passes with 24 of 24.
_FACTS = {}


def report_fact(name: str, value):
    _FACTS[name] = value


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _FACTS:
        return
    import json
    terminalreporter.write_sep("=", "parity facts (measured by the tests above)")
    for k in sorted(_FACTS):
        terminalreporter.write_line(f"{k}: {json.dumps(_FACTS[k])}")
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_facts.json"), "w") as f:
            json.dump(_FACTS, f, indent=1, sort_keys=True)
    except OSError:
        pass
