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

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
