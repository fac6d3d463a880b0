import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rl-collision-avoidance_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"),
          ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library (built on demand; cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    from mrca import _lib
    return _lib.load()
