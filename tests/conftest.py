import faulthandler
import os
import sys

# same runtime configuration as bench.py: read by the HIP runtime when it initialises, i.e. before the first torch.cuda call
# below (the engine keeps three streams busy next to torch's and RCCL's; see klab_multimodalmodel_amd/__init__.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_LOG_FD = None    # the real stderr (pytest's fd capture is suspended during configure): node ids and crash dumps go there
_NATIVE = None    # libklab_mm.so once loaded: its fatal-signal handler prints the running node id and the C backtrace


def pytest_configure(config):
    global _LOG_FD
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        _LOG_FD = os.dup(2)
    except OSError:
        _LOG_FD = None


def _install_native_trace():
    """Order matters: the library's handler first, Python's faulthandler second.  faulthandler then runs first on a fault
    (Python stacks of ALL threads), restores the library's handler and re-raises; the library prints the C backtrace and the
    node id last, so even a log cut at the head names the test."""
    global _NATIVE
    try:
        from klab_multimodalmodel_amd import _lib
        lib = _lib.load()
    except Exception:
        return
    faulthandler.disable()
    lib.klab_segv_trace_install(_LOG_FD if _LOG_FD is not None else -1)
    _NATIVE = lib
    if _LOG_FD is not None:
        faulthandler.enable(file=_LOG_FD, all_threads=True)
    else:
        faulthandler.enable(all_threads=True)


def pytest_collection_modifyitems(config, items):
    # CPU-only hosts: a test marked gpu that gets selected anyway is skipped, never silently passed
    import torch
    if torch.cuda.is_available():
        _install_native_trace()
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_runtest_logstart(nodeid, location):
    # a crash must locate itself: the node id goes to the real stderr (unbuffered, past pytest's capture) before the test
    # starts, and into the native handler's context string
    if _NATIVE is None:
        return
    if _LOG_FD is not None:
        try:
            os.write(_LOG_FD, f"\n[klab-test] {nodeid}\n".encode())
        except OSError:
            pass
    _NATIVE.klab_segv_set_context(nodeid.encode())
