"""Is a tile's extra time per frame a fixed cost or a lower rate?  Tile 0 of G at spp, 2 spp, 4 spp, 8 spp."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
w, h = 1200, 800
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for spp in (25, 50, 100, 200, 400, 800):
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=0, tile_count=G)
        ts = []
        for _ in range(5):
            ctx.render(cam, prm)
            st = ctx.stats()
            ts.append(st.kernel_ms)
        t = statistics.median(ts[1:])
        print(f"G={G} spp {spp:4d}: {t:7.3f} ms  {st.segments / t / 1e6:6.2f} Msegs/ms  ({t / spp * 100:.3f} ms per 100 spp)")
