import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _restore_switches():
    """the library's run-time switches (SATRN_OFF / SATRN_KNOBS / SATRN_PROF / SATRN_TIMING) are process-global: every test starts from, and
    leaves behind, the state it found"""
    import satrn_amd
    snap = satrn_amd.switches.snapshot()
    yield
    satrn_amd.switches.restore(snap)
