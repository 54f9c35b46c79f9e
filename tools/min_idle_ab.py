"""Idle slots a wave of the large-scene kernel must have before it runs a primary pass (no records: C5), interleaved.
usage: min_idle_ab.py [grid_half] [spp]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTIOW_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-rtiow_amd", "librtiow_hip_knobs.so"))  # the RTIOW_DEBUG_* knobs exist in this build only
import vulkan_rtiow_amd as V
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 32
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, grid)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
vals = ["", "8", "16", "24", "40", "48", "64"]
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1)
    res = {v: [] for v in vals}
    for rnd in range(7):
        for v in vals:
            if v:
                os.environ["RTIOW_DEBUG_PASS_MIN_IDLE"] = v
            else:
                os.environ.pop("RTIOW_DEBUG_PASS_MIN_IDLE", None)
            ctx.render(cam, prm)
            ctx.render(cam, prm)
            if rnd:
                res[v].append(ctx.stats().kernel_ms)
    print(f"grid {grid}, {len(sph)} spheres, {spp} spp: " + "  ".join(f"min_idle={v or 'default'}: {statistics.median(res[v]):.2f}" for v in vals))
