import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        from pinocchio_amd import capi
        return capi.lib().brbd_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


def pytest_collection_modifyitems(config, items):
    # GPU tests fail loudly (not skip) when selected with -m gpu on a box without a device or
    # without the built extension; when not selected they are simply deselected by the marker.
    pass
