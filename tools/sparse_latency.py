"""Latency of one sparse iteration against the paths a wave holds, alone on its SIMD and with every wave of the chip in the
same state: the cover scene inside a mirror shell (no path ever ends before max_depth), a frame of `pixels` pixels with P
samples each, and a library built with -DRTIOW_DEBUG_WAVE_POOLS=1 (a wave draws one pool -- here one pixel -- and no more):
every wave that gets a pixel goes straight into its sparse loop with P paths and runs max_depth iterations.
usage: sparse_latency.py lib.so ...   (make OUT=... EXTRA=-DRTIOW_DEBUG_WAVE_POOLS=1)"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_rtiow_amd as V
from importlib import import_module
api = import_module("vulkan-rtiow_amd.api")
libs = [a for a in sys.argv[1:] if a.endswith(".so")] or [os.path.join(os.path.dirname(api.__file__), "librtiow_hip.so")]
sph, mat = V.make_cover_scene(1, 11)
shell_s = np.zeros(1, api.SPHERE_DTYPE); shell_s["radius"] = 60.0
shell_m = np.zeros(1, api.MATERIAL_DTYPE); shell_m["kind"] = api.RT_MAT_METAL; shell_m["albedo"] = 1.0
sph = np.concatenate([sph, shell_s]); mat = np.concatenate([mat, shell_m])
mat["fuzz"] = 0.0  # (a fuzzy reflection below the surface ends its path: none here)
depth = 200
# second scene: the camera INSIDE the big glass ball of the cover scene, whose index is made so large that no ray ever leaves it
# (total internal reflection at every bounce): the fifty-bounce paths that end a real frame, forever
glass = [i for i in range(len(sph)) if abs(sph["radius"][i] - 1.0) < 1e-6 and mat["kind"][i] == api.RT_MAT_DIELECTRIC][0]
mat_glass = mat.copy(); mat_glass["ior"][glass] = 1e6
gc = (float(sph["cx"][glass]), float(sph["cy"][glass]), float(sph["cz"][glass]))
scenes = (("mirror shell", mat, (13, 2, 3), (0, 0, 0), 0.1), ("inside the glass ball", mat_glass, gc, (gc[0] + 1, gc[1] + 0.3, gc[2] + 0.2), 0.0))
for path in libs:
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, at) in api.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = at
    h_ = C.c_void_p()
    assert lib.rtCreate(0, C.byref(h_)) == 0
    assert lib.rtSetScene(h_, sph.ctypes.data, mat.ctypes.data, len(sph)) == 0
    print(os.path.basename(path))
    for scene_name, scene_mat, look_from, look_at, aperture in scenes:
      assert lib.rtSetScene(h_, sph.ctypes.data, scene_mat.ctypes.data, len(sph)) == 0
      print(" " + scene_name)
      for w, h, label in ((4, 2, "8 pixels (lone waves)"), (64, 16, "1024 pixels (a wave per SIMD)"), (64, 64, "4096 pixels (every wave)")):
          cam = V.make_camera(look_from, look_at, (0, 1, 0), 20.0, w / h, aperture, 10.0)
          os.environ["RTIOW_DEBUG_GRID"] = "512"  # (the launch would size the grid for the frame's few samples)
          out = np.zeros((h, w, 4), np.uint8)
          row = []
          for P in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
              ts = []
              for d in (depth, 2 * depth):
                  prm = V.make_params(w, h, spp=P, max_depth=d, seed=1)
                  t = []
                  for _ in range(5):
                      assert lib.rtRender(h_, C.byref(cam), C.byref(prm), out.ctypes.data, w * 4, 0, None) == 0
                      st = V.RtStats(); lib.rtGetStats(h_, C.byref(st))
                      t.append(st.kernel_ms)
                  ts.append(statistics.median(t[1:]))
              row.append((P, (ts[1] - ts[0]) / depth * 1e3, st.segments / (w * h * P), ts[1]))
          print(f"  {label}: us per iteration by paths per wave: " + "  ".join(f"P={P}: {us:.2f} ({t2:.2f} ms)" for P, us, _, t2 in row) +
                f"   (segments per path {row[-1][2]:.0f})")
