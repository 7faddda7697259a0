import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the HIP library is built in-tree and git-ignored: a fresh clone builds it here (hipcc cross-compiles without a GPU)
    from fusion_amd import _lib, _pyhost
    if not os.path.exists(_lib.LIB_PATH) or not os.path.exists(_pyhost.LIB_PATH):
        _lib.build()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as o
    o.build()
    return o
