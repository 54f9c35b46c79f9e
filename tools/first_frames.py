"""Kernel time of the first frames of fresh contexts (the cold end of bench.py's config.first_frame_ms / second_frame_ms): which of
them report chunk costs, which run in an order.  usage: first_frames.py [frames] [contexts]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
prm = V.make_params(w, h, spp=100, max_depth=50, seed=1)
with V.Context(0) as warm:  # the GPU warm before the fresh contexts start
    warm.set_scene(sph, mat)
    for _ in range(10):
        warm.render(cam, prm)
    for r in range(reps):
        with V.Context(0) as ctx:
            ctx.set_scene(sph, mat)
            ts = []
            for k in range(n):
                ctx.render(cam, prm)
                ts.append(ctx.stats().kernel_ms)
            print("fresh context, frames 0.." + str(n - 1) + ": " + " ".join(f"{t:.2f}" for t in ts), flush=True)
