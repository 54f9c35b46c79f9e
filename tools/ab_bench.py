"""Interleaved A/B timing of several builds of librtiow_hip.so in ONE process
(cdna_hip_programming.md 5.4 rule 24).  usage: ab_bench.py libA.so libB.so ... [--spp N] [--rounds R]"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
from importlib import import_module
api = import_module("vulkan-rtiow_amd.api")

args = [a for a in sys.argv[1:] if a.endswith(".so")]
spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 100
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 5
grid = int(sys.argv[sys.argv.index("--grid") + 1]) if "--grid" in sys.argv else 11
tileG = int(sys.argv[sys.argv.index("--tile") + 1]) if "--tile" in sys.argv else 1
kernels = [int(k) for k in sys.argv[sys.argv.index("--kernels") + 1].split(",")] if "--kernels" in sys.argv else [0]
w, h = (1200, 800)
sph, mat = V.make_cover_scene(1, grid)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
libs = []
for path in args:
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, at) in api.SIGNATURES.items():
        if not hasattr(lib, name):
            continue  # an older build
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = at
    h_ = C.c_void_p()
    assert lib.rtCreate(0, C.byref(h_)) == 0
    assert lib.rtSetScene(h_, sph.ctypes.data, mat.ctypes.data, len(sph)) == 0
    libs.append((path, lib, h_))
import numpy as np
out = np.zeros((V.tile_row_count(h, 4, 0, tileG), w, 4), np.uint8)
libs = [(f"{p}:k{k}", lib, h_, k) for p, lib, h_ in libs for k in kernels]
times = {p: [] for p, _, _, _ in libs}
tests = {}
crc = {}
counts = {}
for r in range(rounds + 1):
    for path, lib, h_, k in libs:
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, kernel=k, row_block=4, tile_rank=0, tile_count=tileG)
        assert lib.rtRender(h_, C.byref(cam), C.byref(prm), out.ctypes.data, w * 4, 0, None) == 0
        st = V.RtStats()
        lib.rtGetStats(h_, C.byref(st))
        if r:
            times[path].append(st.kernel_ms)
        crc[path] = int(out.view(np.uint32).sum())
        tests[path] = st.sphere_tests / max(1, st.segments)
        counts[path] = (st.paths, st.segments, st.sphere_tests)
for p in times:
    t = times[p]
    print(f"{os.path.basename(p):40s} median {statistics.median(t):8.3f} ms  min {min(t):8.3f}  frame-sum {crc[p]}  tests/segment {tests[p]:.1f}  paths/segments/tests {counts[p]}")
