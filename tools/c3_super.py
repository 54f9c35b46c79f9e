import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTIOW_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-rtiow_amd", "librtiow_hip_knobs.so"))  # the RTIOW_DEBUG_* knobs exist in this build only
import vulkan_rtiow_amd as V
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
prm = V.make_params(w, h, spp=100, max_depth=50, seed=1)
cases = [("small variant", {}), ("large variant, one level", {"RTIOW_DEBUG_NO_SHADE_LDS": "1"}),
         ("large variant, super level", {"RTIOW_DEBUG_NO_SHADE_LDS": "1", "RTIOW_DEBUG_SUPER_FROM": "24"})]
res = {c[0]: [] for c in cases}
ctxs = {}
for name, env in cases:
    for k in ("RTIOW_DEBUG_NO_SHADE_LDS", "RTIOW_DEBUG_SUPER_FROM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    c = V.Context(0); c.set_scene(sph, mat)   # (the super level is decided in rtSetScene)
    ctxs[name] = (c, env)
for rnd in range(7):
    for name, (c, env) in ctxs.items():
        for k in ("RTIOW_DEBUG_NO_SHADE_LDS", "RTIOW_DEBUG_SUPER_FROM"):
            os.environ.pop(k, None)
        os.environ.update(env)
        c.render(cam, prm); c.render(cam, prm)
        if rnd:
            res[name].append(c.stats().kernel_ms)
        t = c.stats().sphere_tests / c.stats().segments
        res[name + " t"] = t
for name, _ in cases:
    print(f"{name:28s}: {statistics.median(res[name]):.3f} ms  tests/segment {res[name + ' t']:.1f}")
