"""Shared test plumbing.  `-m "not gpu"` runs on the build container (no GPU);
`-m gpu` runs on the MI355X box and calls the HIP path through the C-ABI library."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cross-attention-vit_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fusion_form_of_the_bench_path(request, monkeypatch):
    """GPU tests run at small batches, where the default ("auto") would pick the fusion's literal order (bound by the host there); the path
    the bench runs is the low-rank form, so that is what the parity tests exercise unless a test selects a form itself."""
    if request.node.get_closest_marker("gpu") is not None:
        import xvit.functional as XF
        monkeypatch.setattr(XF, "XATTN_FORM", "lowrank")
    yield
