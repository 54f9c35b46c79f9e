"""Interleaved A/B of several builds on the three-sphere frame (BASELINE config 2: 400x225x100 spp, flat list).
usage: ab_three.py libA.so libB.so ..."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_rtiow_amd as V
from importlib import import_module
api = import_module("vulkan-rtiow_amd.api")
w, h, spp = 400, 225, 100
sph, mat = V.make_three_sphere_scene(False)
cam = V.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, w / h, 0.0, 1.0)
libs = []
for path in [a for a in sys.argv[1:] if a.endswith(".so")]:
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, at) in api.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = at
    h_ = C.c_void_p()
    assert lib.rtCreate(0, C.byref(h_)) == 0
    assert lib.rtSetScene(h_, sph.ctypes.data, mat.ctypes.data, len(sph)) == 0
    libs.append((path, lib, h_))
out = np.zeros((h, w, 4), np.uint8)
times = {p: [] for p, _, _ in libs}
for r in range(8):
    for path, lib, h_ in libs:
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1)
        assert lib.rtRender(h_, C.byref(cam), C.byref(prm), out.ctypes.data, w * 4, 0, None) == 0
        st = V.RtStats()
        lib.rtGetStats(h_, C.byref(st))
        if r:
            times[path].append(st.kernel_ms)
for p, t in times.items():
    print(f"{os.path.basename(p):30s} median {statistics.median(t):7.3f} ms  min {min(t):7.3f}")
