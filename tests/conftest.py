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
def _no_deferred_checks_leak_between_tests():
    """the device-side checks that ride on the next blocking read (eprecon_amd._lib.defer_check) are per process: a test that
    provokes one on purpose, or never reaches a read, must not hand it to the next test"""
    yield
    try:
        from eprecon_amd import _lib
        _lib.take_deferred(None)
    except Exception:  # noqa: BLE001  (library absent: nothing to clear)
        pass
