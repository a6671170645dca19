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


@pytest.fixture(autouse=True, scope="module")
def _networks_of_a_finished_module_are_really_gone():
    """Every test module builds networks whose HIP-graph executables (2D fusion stack, decoder query side) and side streams stay
    alive until the cycle collector runs (the same effect made one bench leg slow the next).  With the round-6
    modules the GPU suite kept enough of them alive for a graph replay late in the run to crash inside the runtime; the modules
    are independent, so each one starts from a collected heap and an empty caching allocator."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:  # noqa: BLE001
        pass
