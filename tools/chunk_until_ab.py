"""How long whole-chunk pools last (RTIOW_DEBUG_CHUNK_UNTIL: pixels left in a queue when the small pools take over) against
frame time, settings interleaved in one process on the knobs build.  usage: chunk_until_ab.py [rounds] [value ...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTIOW_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-rtiow_amd", "librtiow_hip_knobs.so"))  # the RTIOW_DEBUG_* knobs exist in this build only
import vulkan_rtiow_amd as V
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
values = sys.argv[2:] or ["default", "0", "1000", "2000", "4000", "8000"]
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
prm = V.make_params(w, h, spp=100, max_depth=50, seed=1)
ctxs = {}
for v in values:  # one context per setting: the chunk order it uses was made under the same setting
    c = V.Context(0)
    c.set_scene(sph, mat)
    ctxs[v] = c
times = {v: [] for v in values}
for r in range(rounds + 3):
    for v in values:
        if v == "default":
            os.environ.pop("RTIOW_DEBUG_CHUNK_UNTIL", None)
        else:
            os.environ["RTIOW_DEBUG_CHUNK_UNTIL"] = v
        ctxs[v].render(cam, prm)
        if r >= 3:
            times[v].append(ctxs[v].stats().kernel_ms)
for v in values:
    print(f"chunk_until {v:>8s}: median {statistics.median(times[v]):7.3f} ms  min {min(times[v]):7.3f}")
