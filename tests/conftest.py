import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is built once per session if it is not there yet (hipcc cross-compiles
    without a GPU)."""
    lib = os.path.join(ROOT, "freegaussian_amd", "libfgraster.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g

        g.build()
    yield
