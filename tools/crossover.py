"""Flat list against clustered list (with and without the primary pass forced) on small scenes: where the default
should switch (kClusteredFrom in rtiow_kernels.hip).  usage: crossover.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V

w, h, spp = 1200, 800, 100
with V.Context(0) as ctx:
    scenes = [("three", V.make_three_sphere_scene(False), V.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, w / h, 0.0, 1.0))]
    for g in (1, 2, 3, 4, 5):
        scenes.append((f"cover grid {g}", V.make_cover_scene(1, g), V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)))
    for name, (sph, mat), cam in scenes:
        ctx.set_scene(sph, mat)
        row = []
        for k in (2, 3, 4):
            ts = []
            for _ in range(4):
                ctx.render(cam, V.make_params(w, h, spp=spp, max_depth=50, seed=1, kernel=k))
                ts.append(ctx.stats().kernel_ms)
            row.append(statistics.median(ts[1:]))
        print(f"{name:14s} n={len(sph):4d}  flat {row[0]:7.3f} ms  clustered {row[1]:7.3f}  clustered+pass {row[2]:7.3f}", flush=True)
