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


@pytest.fixture(scope="session")
def knobs_ctx(rt):
    """A context on librtiow_hip_knobs.so: the shipped kernels + the RTIOW_DEBUG_* knobs, which the shipped library does not
    read (csrc/rtiow_device.h: debug_knob).  For the parity tests that force a kernel variant through the environment."""
    ctx = rt.Context(0, lib_path=rt.api.KNOBS_LIB_PATH)
    yield ctx
    ctx.close()
