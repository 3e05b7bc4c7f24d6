import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.load()
    return O


@pytest.fixture(scope="session")
def hip_lib():
    """The product library. Fails loudly (no skip, no fallback) when it is missing."""
    import bhr_amd  # noqa: F401
    from bhr_amd import _lib
    return _lib.load()
