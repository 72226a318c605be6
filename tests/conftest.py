import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fe():
    """One FrontEnd (ctx on cuda:0) for the whole GPU session; fails loudly without libvofe.so."""
    from visual_odometry_amd.frontend import FrontEnd
    f = FrontEnd(0)
    yield f
    f.close()



@pytest.fixture
def knob(fe):
    """knob(name, value) sets a VOFE_* switch for one test, knob(name) clears it.  The library reads the switches once
    per context (vofe_create), so the session's context re-reads them after every change and at the end of the test."""
    saved = {}

    def set_(name, value=None):
        if name not in saved:
            saved[name] = os.environ.get(name)
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(value)
        fe.reload_env()

    yield set_
    for name, old in saved.items():
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old
    fe.reload_env()


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
