"""A camera that moves every frame against one that stands still, several builds interleaved in one process: 16 frames of an
orbit around the look-at point (3.75 degrees a frame) by the wall clock around the loop, as bench.py's
config.moving_camera_ms_per_step.  usage: moving_ab.py libA.so libB.so ... [--rounds R] [--step DEGREES]"""
import ctypes as C, math, os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vulkan_rtiow_amd as V
from importlib import import_module
api = import_module("vulkan-rtiow_amd.api")
args = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 5
step = float(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 3.75
w, h, n = 1200, 800, 16
sph, mat = V.make_cover_scene(1, 11)
cam0 = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
r0, a0 = math.hypot(13.0, 3.0), math.atan2(3.0, 13.0)
cams = [V.make_camera((r0 * math.cos(a0 + math.radians(step) * k), 2.0, r0 * math.sin(a0 + math.radians(step) * k)), (0, 0, 0), (0, 1, 0),
                      20.0, w / h, 0.1, 10.0) for k in range(1, n + 1)]
prm = V.make_params(w, h, spp=100, max_depth=50, seed=1)
dev = torch.device("cuda", 0)
bufs = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(n)]
libs = []
for path in args:
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, at) in api.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = at
    h_ = C.c_void_p()
    assert lib.rtCreate(0, C.byref(h_)) == 0
    assert lib.rtSetScene(h_, sph.ctypes.data, mat.ctypes.data, len(sph)) == 0
    libs.append((os.path.basename(path), lib, h_))

def loop(lib, h_, cameras):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k, c in enumerate(cameras):
        assert lib.rtRender(h_, C.byref(c), C.byref(prm), C.c_void_p(bufs[k].data_ptr()), w * 4, 1, None) == 0
    assert lib.rtSynchronize(h_) == 0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(cameras) * 1e3

def settled(lib, h_, c):  # the same view five times over: what it costs once its own chunk order is there
    loop(lib, h_, [c] * 5)
    return loop(lib, h_, [c] * 3)

res = {name: {"static": [], "moving": [], "settled": []} for name, _, _ in libs}
sums = {}
for r in range(rounds + 1):
    for name, lib, h_ in libs:
        loop(lib, h_, [cam0] * 4)
        s = loop(lib, h_, [cam0] * n)
        m = loop(lib, h_, cams)
        sums[name] = int(sum(int(b.to(torch.int64).sum().item()) for b in bufs))
        if r:
            res[name]["static"].append(s)
            res[name]["moving"].append(m)
        if r == 1:
            res[name]["settled"].append(sum(settled(lib, h_, c) for c in cams) / len(cams))
for name in res:
    s, m = statistics.median(res[name]["static"]), statistics.median(res[name]["moving"])
    v = res[name]["settled"][0]
    print(f"{name:24s} start view standing still {s:7.3f} ms/frame   the orbit's 16 views, each standing still {v:7.3f}   orbit ({step} deg/frame) {m:7.3f} ms/frame"
          f"   +{(m / v - 1) * 100:5.1f} % over its own views   orbit frames sum {sums[name]}")
