import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle binding (test infrastructure; builds oracle/liboracle.so if needed)."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def wfst_lib():
    """libwfst_amd.so must exist (built by __graft_entry__.build()); never rebuilt on the GPU box."""
    from rustfst_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from rustfst_amd import build as b
        b.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def gpu_ctx(wfst_lib):
    import rustfst_amd
    ctx = rustfst_amd.Context(0)
    rustfst_amd.set_default_context(ctx)
    return ctx
