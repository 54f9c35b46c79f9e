import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
sph, mat = V.make_cover_scene(1, 11)
def run(ctx, w, h, spp, depth=50, **kw):
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, **kw)
    ts = []
    for _ in range(4):
        ctx.render(cam, prm); st = ctx.stats(); ts.append(st.kernel_ms)
    return statistics.median(ts[1:]), st.segments
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    base, segs0 = run(ctx, 1200, 800, 100)
    print(f"1200x800x100 full        {base:7.2f} ms  {segs0/base/1e6:8.1f} Mseg/ms-1")
    cases = [("1200x800x100 tile 0/8 blk4", (1200, 800, 100), dict(row_block=4, tile_rank=0, tile_count=8)),
             ("1200x800x100 tile 0/2 blk4", (1200, 800, 100), dict(row_block=4, tile_rank=0, tile_count=2))] if "brief" in sys.argv else None
    for name, args, kw in cases or [("1200x800x50 full", (1200, 800, 50), {}), ("1200x800x25 full", (1200, 800, 25), {}),
                           ("1200x800x12 full", (1200, 800, 12), {}),
                           ("1200x400x100 full", (1200, 400, 100), {}), ("600x400x100 full", (600, 400, 100), {}),
                           ("1200x800x100 tile 0/2 blk4", (1200, 800, 100), dict(row_block=4, tile_rank=0, tile_count=2)),
                           ("1200x800x100 tile 0/2 blk400", (1200, 800, 100), dict(row_block=400, tile_rank=0, tile_count=2)),
                           ("1200x800x100 tile 1/2 blk400", (1200, 800, 100), dict(row_block=400, tile_rank=1, tile_count=2)),
                           ("1200x800x100 tile 0/8 blk4", (1200, 800, 100), dict(row_block=4, tile_rank=0, tile_count=8)),
                           ("1200x800x100 d8 full", (1200, 800, 100), dict()),]:
        depth = 8 if "d8" in name else 50
        t, segs = run(ctx, *args, depth=depth, **kw)
        print(f"{name:32s} {t:7.2f} ms  segs {segs/1e6:7.1f}M  rate {segs/t/1e6:7.2f} Mseg/ms  (full-frame rate {segs0/base/1e6:.2f})")
