"""The range of ray origins the cluster boxes are inflated for (scene diagonals from the centre; 2 by default), now that a ray from
beyond it tests enlarged boxes instead of taking every cluster.  usage: range_ab.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTIOW_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-rtiow_amd", "librtiow_hip_knobs.so"))  # the RTIOW_DEBUG_* knobs exist in this build only
import vulkan_rtiow_amd as V
w, h = 1200, 800
for grid, spp, tile in ((32, 64, 1), (20, 64, 1), (14, 64, 1)):
    sph, mat = V.make_cover_scene(1, grid)
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    vals = ["0.3", "0.4", "0.5", "0.6", "0.75", "1.0", "2.0"]
    ctxs = {}
    for v in vals:
        os.environ["RTIOW_DEBUG_RANGE"] = v
        c = V.Context(0); c.set_scene(sph, mat); ctxs[v] = c
    prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4 if tile > 1 else 0, tile_rank=0, tile_count=tile if tile > 1 else 0)
    res = {v: [] for v in vals}; t = {}
    for rnd in range(7):
        for v in vals:
            c = ctxs[v]
            c.render(cam, prm); c.render(cam, prm)
            if rnd: res[v].append(c.stats().kernel_ms)
            t[v] = c.stats().sphere_tests / c.stats().segments
    print(f"grid {grid} spp {spp} tile 1/{tile}: " + "  ".join(f"range {v}: {statistics.median(res[v]):.3f} ms ({t[v]:.1f})" for v in vals), flush=True)
