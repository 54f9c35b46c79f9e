"""Renders the cover frame N times (for profilers).  usage: render_loop.py [N] [G]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=100, max_depth=50, seed=1, row_block=4, tile_rank=0, tile_count=G)
    for _ in range(n):
        ctx.render(cam, prm)
    print(ctx.stats().kernel_ms)
