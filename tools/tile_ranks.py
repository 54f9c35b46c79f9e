"""Kernel time of every tile of an N-way split, several frames each (all values printed): which ranks are slow, and whether
it is the frame or the box.  usage: tile_ranks.py [N] [frames] [rank,rank,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for rank in ([int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else range(G)):
        prm = V.make_params(w, h, spp=100, max_depth=50, seed=1, row_block=4, tile_rank=rank, tile_count=G)
        ts = []
        for _ in range(frames):
            ctx.render(cam, prm)
            ts.append(ctx.stats().kernel_ms)
        print(f"rank {rank}: " + " ".join(f"{t:.3f}" for t in ts) + f"  segs {ctx.stats().segments}", flush=True)
