import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without an AMD GPU skips the gpu-marked tests instead of failing them.
    Only the absence of the kernel driver node counts as "no GPU": on a box that has /dev/kfd the tests run, and a HIP
    runtime that cannot see the device there is a failure, never a silent skip."""
    import pytest
    if os.path.exists("/dev/kfd"):
        return
    markexpr = config.getoption("-m", default="") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return   # the caller asked for the GPU tests by name: let them fail loudly
    skip = pytest.mark.skip(reason="no /dev/kfd on this machine (gpu-marked tests need a real MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
