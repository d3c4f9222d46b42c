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


def pytest_sessionstart(session):
    """GPU runs: a fatal signal inside a runtime library prints its NATIVE call stack (w2c_debug_install_crash_backtrace) before
    Python's faulthandler prints the Python one -- round 4's SIGSEGV left only the Python half in the driver's log."""
    try:
        import torch
        if torch.cuda.is_available():
            from multiagentperception_amd import _native
            fd = 2
            try:                                 # pytest captures fd 2; the faulthandler plugin kept a dup of the real stderr
                from _pytest.faulthandler import fault_handler_stderr_fd_key
                fd = int(session.config.stash[fault_handler_stderr_fd_key])
            except Exception:
                pass
            _native.lib().w2c_debug_install_crash_backtrace(fd)
    except Exception as e:                      # a debug aid must never take the suite down
        print("conftest: crash backtrace handler not installed (%r)" % (e,))
