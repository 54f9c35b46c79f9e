"""What ends a small frame: the same tile with the paths cut at fewer bounces (the work barely changes -- 0.1 % of the paths
reach fifty segments -- so what the time loses is the serial chain of the longest paths).  usage: depth_sweep.py [tile_count] [spp]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
depths = [int(x) for x in os.environ.get("DEPTHS", "50,40,30,20,12,8,5").split(",")]
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    res = {d: [] for d in depths}
    segs = {}
    for rnd in range(6):
        for d in depths:
            prm = V.make_params(w, h, spp=spp, max_depth=d, seed=1, row_block=4, tile_rank=0, tile_count=G)
            for _ in range(3):
                ctx.render(cam, prm)
            if rnd:
                res[d].append(ctx.stats().kernel_ms)
            segs[d] = ctx.stats().segments
    for d in depths:
        print(f"max_depth {d:3d}: median {statistics.median(res[d]):.3f} ms  min {min(res[d]):.3f}  segments {segs[d]}  "
              f"({segs[d] / segs[depths[0]]:.3f} of the work; at the full frame's rate {7.15 * segs[d] / 265678731:.3f} ms)")
