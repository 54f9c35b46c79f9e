"""ctypes binding of oracle/_build/liboracle.so (the CPU checker).  Test infrastructure:
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

import vulkan_rtiow_amd as V  # struct layouts of include/rtiow.h

_VP = C.c_void_p


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_render.restype = C.c_int
        lib.oracle_render.argtypes = [_VP, _VP, C.c_uint32, C.POINTER(V.RtCamera), C.POINTER(V.RtParams),
                                      _VP, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
        lib.oracle_render_ubo.restype = C.c_int
        lib.oracle_render_ubo.argtypes = [C.POINTER(V.RtUbo5), C.c_uint32, _VP, C.c_size_t]
        lib.oracle_tile_row_count.restype = C.c_uint32
        lib.oracle_tile_row_count.argtypes = [C.c_uint32] * 4
        lib.oracle_tile_global_row.restype = C.c_uint32
        lib.oracle_tile_global_row.argtypes = [C.c_uint32] * 4
        lib.oracle_make_camera.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_float] * 4 + [C.POINTER(V.RtCamera)]
        lib.oracle_make_cover_scene.argtypes = [C.c_uint32, C.c_int, _VP, _VP, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.oracle_make_three_sphere_scene.argtypes = [C.c_int, _VP, _VP, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.oracle_write_ppm.argtypes = [C.c_char_p, _VP, C.c_uint32, C.c_uint32, C.c_size_t]
        lib.oracle_arith.argtypes = [C.c_uint32, _VP, _VP, _VP, _VP, C.c_uint32]
        lib.oracle_ubo_from_image.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(V.RtUbo5)]
        lib.oracle_camera_from_ubo.argtypes = [C.POINTER(V.RtUbo5), C.POINTER(V.RtCamera)]

    # -- reference shaders --------------------------------------------------
    def ubo_from_image(self, w, h):
        u = V.RtUbo5()
        assert self.lib.oracle_ubo_from_image(w, h, C.byref(u)) == 0
        return u

    def render_ubo(self, ubo, mode):
        w, h = int(ubo.imageWidth), int(ubo.imageHeight)
        out = np.zeros((h, w, 4), np.uint8)
        rc = self.lib.oracle_render_ubo(C.byref(ubo), mode, out.ctypes.data, w * 4)
        assert rc == 0, rc
        return out

    def camera_from_ubo(self, ubo):
        cam = V.RtCamera()
        assert self.lib.oracle_camera_from_ubo(C.byref(ubo), C.byref(cam)) == 0
        return cam

    # -- PATH -----------------------------------------------------------------
    def render(self, spheres, materials, cam, params, nthreads=0):
        sph = np.ascontiguousarray(spheres, dtype=V.SPHERE_DTYPE)
        mat = np.ascontiguousarray(materials, dtype=V.MATERIAL_DTYPE)
        rows = int(self.lib.oracle_tile_row_count(params.height, params.row_block, params.tile_rank,
                                                  params.tile_count))
        out = np.zeros((rows, params.width, 4), np.uint8)
        segs = C.c_uint64(0)
        rc = self.lib.oracle_render(sph.ctypes.data, mat.ctypes.data, len(sph), C.byref(cam),
                                    C.byref(params), out.ctypes.data, params.width * 4, nthreads,
                                    C.byref(segs))
        assert rc == 0, rc
        return out, segs.value

    def make_camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus):
        cam = V.RtCamera()
        f3 = lambda v: (C.c_float * 3)(*v)
        assert self.lib.oracle_make_camera(f3(lookfrom), f3(lookat), f3(vup), vfov, aspect, aperture,
                                           focus, C.byref(cam)) == 0
        return cam

    def make_cover_scene(self, seed, grid_half):
        cap = (2 * grid_half) ** 2 + 8
        sph = np.zeros(cap, V.SPHERE_DTYPE)
        mat = np.zeros(cap, V.MATERIAL_DTYPE)
        n = C.c_uint32(0)
        assert self.lib.oracle_make_cover_scene(seed, grid_half, sph.ctypes.data, mat.ctypes.data, cap,
                                                C.byref(n)) == 0
        return sph[:n.value].copy(), mat[:n.value].copy()

    def make_three_sphere_scene(self, bubble):
        sph = np.zeros(8, V.SPHERE_DTYPE)
        mat = np.zeros(8, V.MATERIAL_DTYPE)
        n = C.c_uint32(0)
        assert self.lib.oracle_make_three_sphere_scene(int(bubble), sph.ctypes.data, mat.ctypes.data, 8,
                                                       C.byref(n)) == 0
        return sph[:n.value].copy(), mat[:n.value].copy()

    def write_ppm(self, path, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        assert self.lib.oracle_write_ppm(os.fsencode(path), img.ctypes.data, w, h, w * 4) == 0

    def arith(self, op, a, b, c):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        c = np.ascontiguousarray(c, np.float32)
        out = np.zeros_like(a)
        assert self.lib.oracle_arith(op, a.ctypes.data, b.ctypes.data, c.ctypes.data, out.ctypes.data,
                                     a.size) == 0
        return out

    def num_procs(self):
        return int(self.lib.oracle_num_procs())


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load():
    if not os.path.exists(LIB):
        build()  # gcc only; the GPU box normally receives the prebuilt file
    lib = C.CDLL(LIB)
    if not lib.oracle_cpu_ok():
        raise RuntimeError("oracle was built with -mfma but this CPU lacks FMA")
    return Oracle(lib)
