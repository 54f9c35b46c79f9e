"""Renders a BASELINE workload with the block-counting library (tools/blockprof/build.sh) and with the shipped one, checks that the
frames are the same, and leaves the block counters of the LAST instrumented frame in the file RTIOW_BLOCK_DUMP names.
usage: RTIOW_BLOCK_DUMP=gpurun_out/blk.txt python tools/blockprof/run.py [cover|three|cover4096] [spp] [frames] [tile_count]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import vulkan_rtiow_amd as V
from importlib import import_module
api = import_module("vulkan-rtiow_amd.api")

what = sys.argv[1] if len(sys.argv) > 1 else "cover"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tiles = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if what == "three":
    w, h = 400, 225
    sph, mat = V.make_three_sphere_scene(False)
    cam = V.camera_from_ubo(V.ubo_from_image(w, h))  # (bench.py: three_400x225_100spp)
elif what == "cover4096":
    w, h = 3840, 2160
    sph, mat = V.make_cover_scene(1, 32)
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
else:
    w, h = 1200, 800
    sph, mat = V.make_cover_scene(1, 11)
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
out = {}
# (BLOCKPROF_LIB: another instrumented library than the last one built, e.g. a copy kept of each of several builds)
for tag, so in (("shipped", "librtiow_hip.so"), ("blockprof", os.environ.get("BLOCKPROF_LIB", "librtiow_hip_blk.so"))):
    lib = C.CDLL(so if os.path.isabs(so) else os.path.join(ROOT, "vulkan-rtiow_amd", so))
    for name, (res, at) in api.SIGNATURES.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = at
    ctx = C.c_void_p()
    assert lib.rtCreate(0, C.byref(ctx)) == 0
    assert lib.rtSetScene(ctx, sph.ctypes.data, mat.ctypes.data, len(sph)) == 0
    img = np.zeros((V.tile_row_count(h, 4, 0, tiles), w, 4), np.uint8)
    prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=0, tile_count=tiles)
    st = V.RtStats()
    for f in range(frames if tag == "blockprof" else 2):
        assert lib.rtRender(ctx, C.byref(cam), C.byref(prm), img.ctypes.data, w * 4, 0, None) == 0
        lib.rtGetStats(ctx, C.byref(st))
    out[tag] = (img.copy(), st.segments, st.sphere_tests, st.kernel_ms)
    print(f"{tag:10s} {what} {w}x{h}x{spp}: {st.kernel_ms:9.3f} ms, {st.segments} segments, {st.sphere_tests} tests, {st.paths} paths")
    lib.rtDestroy(ctx)
same = np.array_equal(out["shipped"][0], out["blockprof"][0]) and out["shipped"][1] == out["blockprof"][1]
print("frames identical:", same)
sys.exit(0 if same else 1)
