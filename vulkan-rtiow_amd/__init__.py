"""vulkan-rtiow_amd — MI355X-native (gfx950) replacement for the compute-dispatch
hot path of baeng72/Vulkan-RTIOW: hand-written HIP path-tracing kernels behind a
C ABI (include/rtiow.h), this ctypes host binding, and the row-tile / RCCL
framebuffer gather for multi-GPU frames.

The directory name is not a Python identifier; import it through the alias
module `vulkan_rtiow_amd` at the repo root (or importlib.import_module).
"""
from . import api  # noqa: F401
from .api import *  # noqa: F401,F403
from .api import (Context, MultiContext, RtError, chunk_order_selftest_host, cluster_build_host, cone_selftest_host, load_library,  # noqa: F401
                  scene_cluster_selftest_host, camera_is_renderable,
                  multi_selftest_host)  # noqa: F401
