"""The primary pass's records in the large-scene variant: records per wave (RTIOW_DEBUG_PASS_KEEP) against the idle slots a wave
must have before it runs a pass (RTIOW_DEBUG_PASS_MIN_IDLE), cover scenes of 2000-4100 spheres.  usage: keep_ab.py [spp]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTIOW_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-rtiow_amd", "librtiow_hip_knobs.so"))  # the RTIOW_DEBUG_* knobs exist in this build only
import vulkan_rtiow_amd as V
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h = 1200, 800
cases = [("", ""), ("0", ""), ("", "8"), ("", "16"), ("", "32"), ("", "48"), ("0", "16"), ("0", "48")]
for grid in (16, 22, 28, 32):
    sph, mat = V.make_cover_scene(1, grid)
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    with V.Context(0) as ctx:
        ctx.set_scene(sph, mat)
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1)
        res = {c: [] for c in cases}
        for rnd in range(5):
            for c in cases:
                for k, v in zip(("RTIOW_DEBUG_PASS_KEEP", "RTIOW_DEBUG_PASS_MIN_IDLE"), c):
                    if v:
                        os.environ[k] = v
                    else:
                        os.environ.pop(k, None)
                for _ in range(3):
                    ctx.render(cam, prm)
                if rnd:
                    res[c].append(ctx.stats().kernel_ms)
        print(f"grid {grid}: {len(sph)} spheres: " + "  ".join(f"keep={c[0] or 'dflt'},min_idle={c[1] or 'dflt'}: {statistics.median(res[c]):.2f}" for c in cases), flush=True)
