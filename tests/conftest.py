import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture
def run_id():
    import uuid
    return "t" + uuid.uuid4().hex[:10]


# debug aid: DRL_ABORT_BT=1 prints the native backtrace of a thread that aborts / segfaults (scripts/debug/abort_bt.cpp;
# run pytest with -p no:faulthandler so python's handler does not replace it)
if os.environ.get("DRL_ABORT_BT") == "1":
    import ctypes as _ct
    _so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "debug", "abort_bt.so")
    if os.path.exists(_so):
        _ct.CDLL(_so)
