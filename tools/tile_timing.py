"""Kernel time of 1/G of the cover frame (block-cyclic tile 0..G-1) on one GPU: what each rank of a
G-GPU run computes, without the gather.  usage: tile_timing.py [spp]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    full = None
    for G in (1, 2, 4, 8):
        per_rank = []
        for rank in range(G):
            prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, row_block=4, tile_rank=rank, tile_count=G)
            ts = []
            for _ in range(8):
                ctx.render(cam, prm)
                ts.append(ctx.stats().kernel_ms)
            per_rank.append(statistics.median(ts[3:]))
        if G == 1:
            full = per_rank[0]
        print(f"G={G}: per-rank kernel ms {['%.2f' % t for t in per_rank]}  max {max(per_rank):.2f}  "
              f"ideal {full / G:.2f}  compute-only speedup {full / max(per_rank):.2f}x")
