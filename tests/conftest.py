import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu_ctx():
    """One rt_ctx on cuda:0 for the whole GPU session.  Fails (does not skip) when the HIP library or
    the device is missing: the product path has no fallback."""
    import raytrace_amd as R
    ctx = R.Context(0)
    yield ctx
    ctx.close()
