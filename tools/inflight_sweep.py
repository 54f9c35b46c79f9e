"""Paths in flight against the latency of an iteration: the same tile with fewer workgroups (RTIOW_DEBUG_GRID) and
smaller ones (RTIOW_DEBUG_THREADS).  usage: inflight_sweep.py [tile_count] [spp]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTIOW_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-rtiow_amd", "librtiow_hip_knobs.so"))  # the RTIOW_DEBUG_* knobs exist in this build only
import vulkan_rtiow_amd as V
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
cases = [("", ""), ("448", ""), ("384", ""), ("320", ""), ("256", ""), ("", "256"), ("768", "256"), ("512", "256")]
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=0, tile_count=G)
    res = {c: [] for c in cases}
    for rnd in range(7):
        for c in cases:
            for k, v in zip(("RTIOW_DEBUG_GRID", "RTIOW_DEBUG_THREADS"), c):
                if v:
                    os.environ[k] = v
                else:
                    os.environ.pop(k, None)
            for _ in range(3):   # (a new grid is a new frame shape: its first frames have no chunk order yet)
                ctx.render(cam, prm)
            ctx.render(cam, prm)
            if rnd:
                res[c].append(ctx.stats().kernel_ms)
    for c in cases:
        print(f"grid {c[0] or 'default':8s} threads {c[1] or 'default':8s}: median {statistics.median(res[c]):.3f} ms  min {min(res[c]):.3f}")
