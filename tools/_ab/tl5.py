import os, sys
sys.path.insert(0, "/root/repo")
import vulkan_rtiow_amd as V
w, h, spp = 1200, 800, 100
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for rank in (5, 3):
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=rank, tile_count=8)
        for _ in range(4):
            ctx.render(cam, prm)
        sys.stderr.write(f"--- rank {rank}\n")
        st = ctx.stats()
        print(f"rank {rank}: {st.kernel_ms:.3f} ms")
