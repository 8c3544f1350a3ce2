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


# ---- the parity report: every "[label] ..." line a test prints — how many samples were bounded against the oracle,
# which share of them the tolerance model declared ill-conditioned (compared on finiteness / flags only), how many flags differed, how
# many trees a certificate covered — is collected from the captured output of PASSING tests too and printed at the end of the run, so
# that it shows in `pytest -q` logs (GPUTEST records) and not only under `-s`.
_PARITY_LINES = []


def pytest_runtest_logreport(report):
    if report.when != "call":
        return
    for line in (report.capstdout or "").splitlines():
        if line.startswith("[") and "]" in line:
            _PARITY_LINES.append((report.nodeid.split("::", 1)[-1], line.strip()))


def pytest_terminal_summary(terminalreporter):
    if not _PARITY_LINES:
        return
    terminalreporter.section("parity report (samples bounded / ill-conditioned share / flags, per test)")
    for nodeid, line in _PARITY_LINES:
        terminalreporter.write_line(f"{nodeid}: {line[:400]}")
