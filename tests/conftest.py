import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def pytest_sessionstart(session):
    """The product library, the host mirror and its executables are built in-tree and kept out of git.  `make` runs
    every time: it is a no-op when everything is up to date, and it means the tests can never pass against a
    binary older than the sources (hipcc cross-compiles gfx950 without a GPU; ~40 s from scratch)."""
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    O.set_math(0)
    return O
