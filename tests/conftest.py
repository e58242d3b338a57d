import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    oracle.set_num_threads(min(oracle.usable_cpus(), 16))   # the GPU box shares a 256-CPU host
    return oracle


@pytest.fixture
def hooks():
    """Per-context test hooks (rslf_ctx_set_debug) on the default context of cuda:0, reset after the test."""
    from remotesensingproject_amd import depth as rs
    ctx = rs.default_context(0)
    yield ctx.set_debug
    ctx.reset_debug()
