import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_bind
    return oracle_bind.load()


@pytest.fixture(scope="session")
def rt():
    import vulkan_rtiow_amd as V
    V.load_library()
    return V


@pytest.fixture(scope="session")
def gpu_ctx(rt):
    ctx = rt.Context(0)  # raises (does not skip) when the HIP path is unusable
    yield ctx
    ctx.close()
