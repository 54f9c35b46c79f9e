"""ctypes binding of the C ABI declared in include/rtiow.h.

This is the Python host side of the drop-in boundary: it mirrors the reference's
resource/dispatch interface (camera UBO in, RGBA8 image out —
RTCHAP06/main.cpp:109-157 and :313-325) one call for one call.  There is no
CPU fallback here: if librtiow_hip.so is missing, or no GPU is usable, the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTIOW_LIB lets a diagnostic build of the same ABI (e.g. -DRTIOW_DEBUG_COUNTERS) stand in
LIB_PATH = os.environ.get("RTIOW_LIB") or os.path.join(_HERE, "librtiow_hip.so")
# The shipped kernels + the RTIOW_DEBUG_* tuning / test knobs, which the shipped library does not read (csrc/rtiow_device.h:
# debug_knob; `make -C csrc knobs`): Context(lib_path=KNOBS_LIB_PATH) for the tools and tests that force a kernel variant.
KNOBS_LIB_PATH = os.path.join(_HERE, "librtiow_hip_knobs.so")

# ---- enums (include/rtiow.h) -------------------------------------------------
RT_OK = 0
RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_NOMEM, RT_ERR_STATE, RT_ERR_IO = 1, 2, 3, 4, 5, 6
RT_MODE_CH05, RT_MODE_CH06, RT_MODE_PATH = 5, 6, 13
RT_QUANT_UNORM8, RT_QUANT_BOOK = 0, 1
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC = 0, 1, 2
KERNEL_DEFAULT, KERNEL_PIXEL, KERNEL_PERSISTENT, KERNEL_CLUSTERED, KERNEL_CLUSTERED_PASS = 0, 1, 2, 3, 4


class RtUbo5(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("imageWidth", "imageHeight", "viewportWidth", "viewportHeight", "focalLength")]


class RtSphere(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("radius", C.c_float)]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", C.c_float * 3), ("fuzz", C.c_float),
                ("ior", C.c_float), ("pad", C.c_uint32 * 2)]


class RtCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float)]


class RtParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("width", "height", "spp", "max_depth", "seed", "mode", "quantiser", "chunk_spp",
                 "row_block", "tile_rank", "tile_count", "kernel", "sample_offset", "accumulate")]


class RtSceneStats(C.Structure):
    _fields_ = [("scene_build_ms", C.c_double), ("last_cluster_build_ms", C.c_double), ("range_diags", C.c_double),
                ("base_range_diags", C.c_double), ("cluster_builds", C.c_uint32), ("n_clusters", C.c_uint32),
                ("n_super", C.c_uint32), ("n_large", C.c_uint32), ("flat_axis", C.c_uint32), ("n_spheres", C.c_uint32)]


class RtStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("paths", C.c_uint64), ("segments", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("bytes_written", C.c_uint64),
                ("rows_rendered", C.c_uint32), ("n_spheres", C.c_uint32), ("debug", C.c_uint64 * 8),
                ("shader_clock_mhz", C.c_uint32), ("reserved", C.c_uint32)]


SPHERE_DTYPE = np.dtype([("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4"), ("radius", "<f4")])
MATERIAL_DTYPE = np.dtype([("kind", "<u4"), ("albedo", "<f4", (3,)), ("fuzz", "<f4"),
                           ("ior", "<f4"), ("pad", "<u4", (2,))])
assert SPHERE_DTYPE.itemsize == 16 and MATERIAL_DTYPE.itemsize == 32
assert C.sizeof(RtUbo5) == 20 and C.sizeof(RtMaterial) == 32 and C.sizeof(RtCamera) == 88

# every symbol include/rtiow.h declares, with its signature
_VP = C.c_void_p
SIGNATURES = {
    "rtCreate": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "rtDestroy": (C.c_int, [_VP]),
    "rtGetLastError": (C.c_char_p, [_VP]),
    "rtAbiVersion": (C.c_int, []),
    "rtSetScene": (C.c_int, [_VP, _VP, _VP, C.c_uint32]),
    "rtRender": (C.c_int, [_VP, C.POINTER(RtCamera), C.POINTER(RtParams), _VP, C.c_size_t, C.c_int, _VP]),
    "rtRenderUbo": (C.c_int, [_VP, C.POINTER(RtUbo5), C.c_uint32, _VP, C.c_size_t, C.c_int, _VP]),
    "rtGetStats": (C.c_int, [_VP, C.POINTER(RtStats)]),
    "rtSynchronize": (C.c_int, [_VP]),
    "rtGetLastKernel": (C.c_int, [_VP, C.POINTER(C.c_uint32)]),
    "rtSelfTestArith": (C.c_int, [_VP, C.c_uint32, _VP, _VP, _VP, _VP, C.c_uint32]),
    "rtSelfTestChSkySteps": (C.c_int, [_VP, C.c_float, C.c_float, _VP, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rtSelfTestUnaryScan": (C.c_int, [_VP, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_uint64), _VP, C.c_uint32]),
    "rtUboFromImage": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(RtUbo5)]),
    "rtCameraFromUbo": (C.c_int, [C.POINTER(RtUbo5), C.POINTER(RtCamera)]),
    "rtMakeCamera": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                               C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(RtCamera)]),
    "rtMakeCoverScene": (C.c_int, [C.c_uint32, C.c_int, _VP, _VP, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rtMakeThreeSphereScene": (C.c_int, [C.c_int, _VP, _VP, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rtTileRowCount": (C.c_uint32, [C.c_uint32] * 4),
    "rtTileGlobalRow": (C.c_uint32, [C.c_uint32] * 4),
    "rtWritePPM": (C.c_int, [C.c_char_p, _VP, C.c_uint32, C.c_uint32, C.c_size_t]),
    "rtWritePNG": (C.c_int, [C.c_char_p, _VP, C.c_uint32, C.c_uint32, C.c_size_t]),
    "rtCreateMulti": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_VP)]),
    "rtDestroyMulti": (C.c_int, [_VP]),
    "rtMultiSetScene": (C.c_int, [_VP, _VP, _VP, C.c_uint32]),
    "rtMultiRender": (C.c_int, [_VP, C.POINTER(RtCamera), C.POINTER(RtParams), _VP, C.c_size_t, C.c_int]),
    "rtMultiSynchronize": (C.c_int, [_VP]),
    "rtMultiGetStats": (C.c_int, [_VP, C.c_int, C.POINTER(RtStats), C.POINTER(C.c_double)]),
    "rtMultiDeviceCount": (C.c_int, [_VP]),
    "rtMultiTransport": (C.c_char_p, [_VP]),
    "rtMultiGetLastError": (C.c_char_p, [_VP]),
    "rtMultiSelfTestHost": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _VP]),
    "rtChunkOrderSelfTestHost": (C.c_int, [C.c_uint32, _VP, _VP]),
    "rtClusterBuildHost": (C.c_int, [_VP, C.c_uint32, C.c_float, _VP, _VP, C.c_uint32, _VP, _VP, _VP, C.c_uint32, _VP, _VP, _VP, _VP]),
    "rtSceneClusterSelfTestHost": (C.c_int, [_VP, C.c_uint32, _VP, _VP, _VP]),
    "rtCameraIsRenderable": (C.c_int, [_VP]),
    "rtGetSceneStats": (C.c_int, [_VP, _VP]),
    "rtConeSelfTestHost": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _VP, C.c_float, _VP, C.c_uint32,
                                     _VP, C.c_uint32, _VP, _VP]),
}

_lib: Optional[C.CDLL] = None


class RtError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        super().__init__(f"{where} failed with status {code}" + (f": {detail}" if detail else ""))
        self.code = code


def _share_torch_hip_runtime() -> None:
    """PyTorch wheels carry their own libamdhip64.so (same SONAME as /opt/rocm's, different file) and
    ask for it by the name `libamdhip64.so`.  If this library is loaded first it binds /opt/rocm's
    copy, torch later loads its own as a second HIP runtime in the process, and that one finds no GPU.
    Loading torch's copy first (by path, without importing torch) lets both bind the same runtime,
    whichever of the two is imported first."""
    if "torch" in sys.modules:
        return  # already loaded: our libamdhip64.so.7 resolves to torch's copy by SONAME
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # no torch, or an unusual layout: /opt/rocm's runtime is used
        pass


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Loads librtiow_hip.so and types every exported entry point.  Raises if it is absent."""
    global _lib
    if _lib is not None and path == LIB_PATH:
        return _lib
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the render path.")
    _share_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.rtAbiVersion() != 4:
        raise ImportError("librtiow_hip.so has an unexpected ABI version")
    if path == LIB_PATH:
        _lib = lib
    return lib


def _f3(v: Sequence[float]):
    return (C.c_float * 3)(*[float(x) for x in v])


# ---- host-side helpers (no GPU) ------------------------------------------------
def ubo_from_image(width: int, height: int) -> RtUbo5:
    ubo = RtUbo5()
    _check(None, load_library().rtUboFromImage(width, height, C.byref(ubo)), "rtUboFromImage")
    return ubo


def camera_from_ubo(ubo: RtUbo5) -> RtCamera:
    cam = RtCamera()
    _check(None, load_library().rtCameraFromUbo(C.byref(ubo), C.byref(cam)), "rtCameraFromUbo")
    return cam


def make_camera(lookfrom, lookat, vup, vfov_deg, aspect, aperture, focus_dist) -> RtCamera:
    cam = RtCamera()
    _check(None, load_library().rtMakeCamera(_f3(lookfrom), _f3(lookat), _f3(vup), vfov_deg, aspect,
                                             aperture, focus_dist, C.byref(cam)), "rtMakeCamera")
    return cam


def make_cover_scene(seed: int = 1, grid_half: int = 11) -> Tuple[np.ndarray, np.ndarray]:
    cap = (2 * grid_half) ** 2 + 8
    sph = np.zeros(cap, SPHERE_DTYPE)
    mat = np.zeros(cap, MATERIAL_DTYPE)
    n = C.c_uint32(0)
    _check(None, load_library().rtMakeCoverScene(seed, grid_half, sph.ctypes.data, mat.ctypes.data,
                                                 cap, C.byref(n)), "rtMakeCoverScene")
    return sph[:n.value].copy(), mat[:n.value].copy()


def make_three_sphere_scene(with_bubble: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    sph = np.zeros(8, SPHERE_DTYPE)
    mat = np.zeros(8, MATERIAL_DTYPE)
    n = C.c_uint32(0)
    _check(None, load_library().rtMakeThreeSphereScene(int(with_bubble), sph.ctypes.data,
                                                       mat.ctypes.data, 8, C.byref(n)),
           "rtMakeThreeSphereScene")
    return sph[:n.value].copy(), mat[:n.value].copy()


def tile_row_count(height: int, row_block: int, rank: int, count: int) -> int:
    return int(load_library().rtTileRowCount(height, row_block, rank, count))


def tile_global_row(local_row: int, row_block: int, rank: int, count: int) -> int:
    return int(load_library().rtTileGlobalRow(local_row, row_block, rank, count))


def write_ppm(path: str, rgba8: np.ndarray) -> None:
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w = img.shape[:2]
    _check(None, load_library().rtWritePPM(os.fsencode(path), img.ctypes.data, w, h, w * 4), "rtWritePPM")


def write_png(path: str, rgba8: np.ndarray) -> None:
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w = img.shape[:2]
    _check(None, load_library().rtWritePNG(os.fsencode(path), img.ctypes.data, w, h, w * 4), "rtWritePNG")


def _check(ctx, code: int, where: str) -> None:
    if code != RT_OK:
        lib = load_library()
        msg = lib.rtGetLastError(ctx)
        raise RtError(code, where, msg.decode() if msg else "")


def make_params(width, height, spp=1, max_depth=50, seed=1, mode=RT_MODE_PATH,
                quantiser=RT_QUANT_BOOK, chunk_spp=0, row_block=0, tile_rank=0, tile_count=0,
                kernel=KERNEL_DEFAULT, sample_offset=0, accumulate=0) -> RtParams:
    return RtParams(width, height, spp, max_depth, seed, mode, quantiser, chunk_spp, row_block,
                    tile_rank, tile_count, kernel, sample_offset, accumulate)


class Context:
    """One GPU's render context (rtCreate .. rtDestroy)."""

    def __init__(self, device_id: int = 0, lib_path: Optional[str] = None):
        self._lib = load_library(lib_path) if lib_path else load_library()
        self._h = _VP()
        code = self._lib.rtCreate(device_id, C.byref(self._h))
        if code != RT_OK:
            msg = self._lib.rtGetLastError(None)
            raise RtError(code, "rtCreate", msg.decode() if msg else "")
        self.device_id = device_id

    def close(self) -> None:
        if self._h:
            self._lib.rtDestroy(self._h)
            self._h = _VP()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, spheres: np.ndarray, materials: np.ndarray) -> None:
        sph = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        mat = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        if len(sph) != len(mat):
            raise ValueError("spheres and materials differ in length")
        _check(self._h, self._lib.rtSetScene(self._h, sph.ctypes.data, mat.ctypes.data, len(sph)),
               "rtSetScene")

    def render(self, cam: Optional[RtCamera], params: RtParams) -> np.ndarray:
        """Renders this tile's rows into a host array [rows, width, 4] (row 0 = scene bottom)."""
        rows = tile_row_count(params.height, params.row_block, params.tile_rank, params.tile_count)
        out = np.zeros((rows, params.width, 4), np.uint8)
        dst = out.ctypes.data if rows else np.zeros(4, np.uint8).ctypes.data
        _check(self._h, self._lib.rtRender(self._h, C.byref(cam) if cam is not None else None,
                                           C.byref(params), dst, params.width * 4, 0, None),
               "rtRender")
        return out

    def render_device(self, cam: Optional[RtCamera], params: RtParams, dst_ptr: int, pitch: int,
                      stream: int = 0) -> None:
        """Enqueues the render into device memory `dst_ptr` on hipStream_t `stream` (no host sync)."""
        _check(self._h, self._lib.rtRender(self._h, C.byref(cam) if cam is not None else None,
                                           C.byref(params), _VP(dst_ptr), pitch, 1,
                                           _VP(stream) if stream else None), "rtRender")

    def render_ubo(self, ubo: RtUbo5, mode: int) -> np.ndarray:
        w, h = int(ubo.imageWidth), int(ubo.imageHeight)
        out = np.zeros((h, w, 4), np.uint8)
        _check(self._h, self._lib.rtRenderUbo(self._h, C.byref(ubo), mode, out.ctypes.data, w * 4, 0,
                                              None), "rtRenderUbo")
        return out

    def render_ubo_device(self, ubo: RtUbo5, mode: int, dst_ptr: int, pitch: int, stream: int = 0) -> None:
        """The reference's dispatch (20-byte UBO in, RGBA8 image out) into device memory, no host sync."""
        _check(self._h, self._lib.rtRenderUbo(self._h, C.byref(ubo), mode, _VP(dst_ptr), pitch, 1,
                                              _VP(stream) if stream else None), "rtRenderUbo")

    def synchronize(self) -> None:
        _check(self._h, self._lib.rtSynchronize(self._h), "rtSynchronize")

    def stats(self) -> RtStats:
        st = RtStats()
        _check(self._h, self._lib.rtGetStats(self._h, C.byref(st)), "rtGetStats")
        return st

    def scene_stats(self) -> RtSceneStats:
        st = RtSceneStats()
        _check(self._h, self._lib.rtGetSceneStats(self._h, C.byref(st)), "rtGetSceneStats")
        return st

    def last_kernel(self) -> int:
        """Kernel variant the last PATH render launched (KERNEL_PIXEL / _PERSISTENT / _CLUSTERED)."""
        k = C.c_uint32(0)
        _check(self._h, self._lib.rtGetLastKernel(self._h, C.byref(k)), "rtGetLastKernel")
        return int(k.value)

    def selftest_arith(self, op: int, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        c = np.ascontiguousarray(c, np.float32)
        out = np.zeros_like(a)
        _check(self._h, self._lib.rtSelfTestArith(self._h, op, a.ctypes.data, b.ctypes.data,
                                                  c.ctypes.data, out.ctypes.data, a.size),
               "rtSelfTestArith")
        return out

    def selftest_unary_scan(self, fn: int, lo: float, hi: float, cap: int = 16):
        """rtSelfTestUnaryScan: (number of floats in [lo, hi] on which the kernels' square root (fn 0) / reciprocal (fn 1) differs from
        sqrtf / 1.0f / x, the first few of them)."""
        bad = C.c_uint64(0)
        first = np.zeros(max(cap, 1), np.uint32)
        _check(self._h, self._lib.rtSelfTestUnaryScan(self._h, fn, lo, hi, C.byref(bad), first.ctypes.data, cap), "rtSelfTestUnaryScan")
        return int(bad.value), first[: min(cap, int(bad.value))].view(np.float32)

    def selftest_ch_sky_steps(self, lo: float, hi: float, cap: int = 4096) -> np.ndarray:
        """rtSelfTestChSkySteps: the floats in [lo, hi] where the CH shaders' sky colour changes, sorted, as a structured array
        (unit_y, before, after)."""
        out = np.zeros(cap, dtype=[("unit_y", np.float32), ("before", np.uint32), ("after", np.uint32)])
        n = C.c_uint32(0)
        _check(self._h, self._lib.rtSelfTestChSkySteps(self._h, lo, hi, out.ctypes.data, cap, C.byref(n)), "rtSelfTestChSkySteps")
        if n.value > cap:
            raise RtError(RT_ERR_INVALID, "rtSelfTestChSkySteps", f"{n.value} steps, room for {cap}")
        return np.sort(out[: n.value], order="unit_y")


class MultiContext:
    """Several GPUs of one node driven from this process (rtCreateMulti .. rtDestroyMulti): block-cyclic
    row tiles, one RCCL gather to device_ids[0], de-interleave there.  A list that repeats a device is the
    one-GPU rehearsal of the N-tile path (tiles moved by hipMemcpyPeerAsync)."""

    def __init__(self, device_ids: Sequence[int]):
        self._lib = load_library()
        self._h = _VP()
        ids = (C.c_int * len(device_ids))(*device_ids)
        code = self._lib.rtCreateMulti(ids, len(device_ids), C.byref(self._h))
        if code != RT_OK:
            msg = self._lib.rtMultiGetLastError(None)
            raise RtError(code, "rtCreateMulti", msg.decode() if msg else "")
        self.device_ids = list(device_ids)

    def _check(self, code: int, where: str) -> None:
        if code != RT_OK:
            msg = self._lib.rtMultiGetLastError(self._h)
            raise RtError(code, where, msg.decode() if msg else "")

    def close(self) -> None:
        if self._h:
            self._lib.rtDestroyMulti(self._h)
            self._h = _VP()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def transport(self) -> str:
        return self._lib.rtMultiTransport(self._h).decode()

    def set_scene(self, spheres: np.ndarray, materials: np.ndarray) -> None:
        spheres = np.ascontiguousarray(spheres, SPHERE_DTYPE)
        materials = np.ascontiguousarray(materials, MATERIAL_DTYPE)
        self._check(self._lib.rtMultiSetScene(self._h, spheres.ctypes.data, materials.ctypes.data, len(spheres)),
                    "rtMultiSetScene")

    def render(self, cam: RtCamera, params: RtParams) -> np.ndarray:
        """The assembled frame as a host array [height, width, 4] (row 0 = scene bottom)."""
        out = np.zeros((params.height, params.width, 4), np.uint8)
        self._check(self._lib.rtMultiRender(self._h, C.byref(cam), C.byref(params), out.ctypes.data,
                                            params.width * 4, 0), "rtMultiRender")
        return out

    def render_device(self, cam: RtCamera, params: RtParams, dst_ptr: int, pitch: int) -> None:
        """Enqueues the frame into device memory on device_ids[0]; synchronize() waits for it."""
        self._check(self._lib.rtMultiRender(self._h, C.byref(cam), C.byref(params), _VP(dst_ptr), pitch, 1),
                    "rtMultiRender")

    def synchronize(self) -> None:
        self._check(self._lib.rtMultiSynchronize(self._h), "rtMultiSynchronize")

    def stats(self, device_index: int = 0) -> Tuple[RtStats, float]:
        st = RtStats()
        ms = C.c_double(0.0)
        self._check(self._lib.rtMultiGetStats(self._h, device_index, C.byref(st), C.byref(ms)), "rtMultiGetStats")
        return st, float(ms.value)


def multi_selftest_host(full: np.ndarray, row_block: int, n_tiles: int) -> np.ndarray:
    """rtMultiSelfTestHost: cuts `full` ([height, width] uint32) into n_tiles tiles, moves them through the
    host stand-in for the communicator and reassembles them (no GPU involved)."""
    lib = load_library()
    full = np.ascontiguousarray(full, np.uint32)
    out = np.empty_like(full)
    code = lib.rtMultiSelfTestHost(full.ctypes.data, full.shape[1], full.shape[0], row_block, n_tiles, out.ctypes.data)
    if code != RT_OK:
        raise RtError(code, "rtMultiSelfTestHost", (lib.rtMultiGetLastError(None) or b"").decode())
    return out


def cluster_build_host(spheres: np.ndarray, range_diags: float = 2.0) -> dict:
    """rtClusterBuildHost: the two-level list rtSetScene builds (no GPU involved) -- boxes [n, 6] of the clusters then the
    super-clusters, slot_index [n_slots] (0xFFFFFFFF: padding), n_large_slots, flat_axis (3: none), flat_interval
    (centre, half extent) and flat_boxes [n, 4]."""
    lib = load_library()
    spheres = np.ascontiguousarray(spheres, SPHERE_DTYPE)
    cap = len(spheres) // 8 + 64
    slot_cap = 2 * len(spheres) + 1024
    boxes = np.zeros((cap, 6), np.float32)
    flat = np.zeros((cap, 4), np.float32)
    slots = np.zeros(slot_cap, np.uint32)
    nc, ns, nslots, nlarge, axis = (C.c_uint32(0) for _ in range(5))
    interval = (C.c_float * 2)()
    code = lib.rtClusterBuildHost(spheres.ctypes.data, len(spheres), float(range_diags), boxes.ctypes.data, flat.ctypes.data, cap,
                                  C.byref(nc), C.byref(ns), slots.ctypes.data, slot_cap, C.byref(nslots), C.byref(nlarge),
                                  C.byref(axis), interval)
    if code != RT_OK:
        raise RtError(code, "rtClusterBuildHost", (lib.rtGetLastError(None) or b"").decode())
    n = nc.value + ns.value
    assert n <= cap and nslots.value <= slot_cap
    return {"n_clusters": nc.value, "n_super": ns.value, "boxes": boxes[:n].copy(), "flat_boxes": flat[:n].copy(),
            "slot_index": slots[:nslots.value].copy(), "n_large_slots": nlarge.value, "flat_axis": axis.value,
            "flat_interval": (float(interval[0]), float(interval[1]))}


def camera_is_renderable(cam: "RtCamera") -> bool:
    """rtCameraIsRenderable: rtRender's precondition on the camera (ray lengths within [2^-30, 2^40], image plane not degenerate)."""
    return bool(load_library().rtCameraIsRenderable(C.byref(cam)))


def scene_cluster_selftest_host(spheres: np.ndarray) -> dict:
    """rtSceneClusterSelfTestHost: rtSetScene's own choice of range and levels, run on the CPU -- how many cluster builds it
    took, the range (scene diagonals) and the super-clusters of the lists it would upload."""
    lib = load_library()
    spheres = np.ascontiguousarray(spheres)
    builds, n_super, rng = C.c_uint32(0), C.c_uint32(0), C.c_double(0.0)
    code = lib.rtSceneClusterSelfTestHost(spheres.ctypes.data, len(spheres), C.byref(builds), C.byref(rng), C.byref(n_super))
    if code != RT_OK:
        raise RtError(code, "rtSceneClusterSelfTestHost")
    return {"builds": builds.value, "range_diags": rng.value, "n_super": n_super.value}


def chunk_order_selftest_host(n_chunks: int) -> Tuple[int, int]:
    """rtChunkOrderSelfTestHost: (words rtRender allocates for the chunk order of n_chunks chunks, highest word the
    kernels index).  No GPU involved."""
    lib = load_library()
    words, hi = C.c_uint32(0), C.c_uint32(0)
    code = lib.rtChunkOrderSelfTestHost(n_chunks, C.byref(words), C.byref(hi))
    if code != RT_OK:
        raise RtError(code, "rtChunkOrderSelfTestHost", (lib.rtGetLastError(None) or b"").decode())
    return int(words.value), int(hi.value)


def cone_selftest_host(cam: RtCamera, width: int, height: int, pix_lo: int, pix_hi: int, range_center, range_rmax: float,
                       spheres: np.ndarray, boxes: np.ndarray):
    """rtConeSelfTestHost: the primary pass's cull (the kernels' own functions, compiled for the host) for the span of
    pixels pix_lo..pix_hi of one row: (cull is on?, per-sphere reach flags, per-box reach flags).  No GPU involved."""
    lib = load_library()
    spheres = np.ascontiguousarray(spheres, SPHERE_DTYPE)
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    centre = np.ascontiguousarray(range_center, np.float32)
    sr = np.zeros(len(spheres), np.uint8)
    br = np.zeros(len(boxes), np.uint8)
    code = lib.rtConeSelfTestHost(C.byref(cam), width, height, pix_lo, pix_hi, centre.ctypes.data, float(range_rmax),
                                  spheres.ctypes.data, len(spheres), boxes.ctypes.data, len(boxes), sr.ctypes.data, br.ctypes.data)
    if code < 0:
        raise RtError(-code, "rtConeSelfTestHost", (lib.rtGetLastError(None) or b"").decode())
    return bool(code), sr.astype(bool), br.astype(bool)
