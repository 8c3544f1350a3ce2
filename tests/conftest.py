import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_cases():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as fh:
        return json.load(fh)["cases"]


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a visible MI355X: on a box without one they are SKIPPED (with the reason), not errors.
    On a GPU box nothing is skipped — a missing libde_hip.so then fails loudly inside the tests' `api` fixtures
    (there is no CPU fallback to fall back to)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    if have_gpu or os.environ.get("DE_REQUIRE_GPU") == "1":
        return
    skip = pytest.mark.skip(reason="no GPU visible (torch.cuda.is_available() is False); set DE_REQUIRE_GPU=1 to fail instead")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
