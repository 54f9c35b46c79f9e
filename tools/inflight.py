"""Throughput of a frame loop with F frames in flight (F contexts on F streams) on one GPU:
full frame and a 1/8 tile.  usage: inflight.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vulkan_rtiow_amd as V

w, h, spp = 1200, 800, 100
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
for G in (1, 8):
    for F in (1, 2, 3, 4, 6):
        ctxs = [V.Context(0) for _ in range(F)]
        streams = [torch.cuda.Stream() for _ in range(F)]
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=0, tile_count=G)
        rows = V.tile_row_count(h, 4, 0, G)
        bufs = [torch.zeros((rows, w), dtype=torch.int32, device="cuda") for _ in range(F)]
        for c in ctxs:
            c.set_scene(sph, mat)
        def frame(k):
            i = k % F
            ctxs[i].render_device(cam, prm, bufs[i].data_ptr(), w * 4, streams[i].cuda_stream)
        for k in range(2 * F):
            frame(k)
        torch.cuda.synchronize()
        n = 24
        t0 = time.perf_counter()
        for k in range(n):
            frame(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        print(f"tile 1/{G}: {F} frame(s) in flight: {dt:.2f} ms/frame")
        for c in ctxs:
            c.close()
