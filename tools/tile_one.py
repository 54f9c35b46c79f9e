import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h, spp = 1200, 800, 100
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=0, tile_count=G)
    for _ in range(4):
        ctx.render(cam, prm)
    print(ctx.stats().kernel_ms)
