"""pytest configuration: the `gpu` marker and import paths."""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class _Knobs:
    """Schedule knobs of libfrcnn_hip.so with monkeypatch's setenv / delenv interface.  The library reads its FRCNN_*
    environment variables once at load time; afterwards a schedule is forced through frcnn_debug_set_schedule."""

    def __init__(self):
        from tf_faster_rcnn_amd import _lib
        self._lib = _lib
        self._touched = set()

    def setenv(self, name, value):
        self._touched.add(name)
        self._lib.set_schedule(name, int(value))

    def delenv(self, name, raising=False):
        self._touched.add(name)
        self._lib.set_schedule(name, None)

    def restore(self):
        for name in self._touched:
            self._lib.set_schedule(name, None)


@pytest.fixture
def knobs():
    k = _Knobs()
    yield k
    k.restore()
