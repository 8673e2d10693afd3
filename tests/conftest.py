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
    """The product library, the host mirror and its executables are built in-tree and kept out of git: build them
    when a fresh checkout has none (hipcc cross-compiles gfx950 without a GPU; ~40 s).  Nothing is rebuilt otherwise."""
    pkg = os.path.join(ROOT, "path_tracer_ocaml_amd")
    built = [os.path.join(pkg, n) for n in ("libptx_hip.so", "libpt_host.so", "shirley_spheres", "cornell_box", "ganesha")]
    if not all(os.path.exists(b) for b in built):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    O.set_math(0)
    return O
