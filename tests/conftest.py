import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _restore_library_tuning(request):
    """cpt_set_tuning is process-global (A/B switches): whatever a GPU test flips -- also one that fails half way -- is put
    back to the defaults before the next test runs (a failing test used to leave the library mis-tuned)."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        try:
            from cpt_amd import _lib as L
            L.lib().cpt_set_tuning(-1, 0)
        except Exception:
            pass
