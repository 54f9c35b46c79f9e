"""Kernel time of the cover frame at low samples per pixel (the progressive / interactive regime:
one dispatch of a few spp per displayed frame, accumulated)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for spp in (1, 2, 4, 8, 16, 32, 100):
        for acc in (0, 1):
            ts = []
            for f in range(6):
                prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, accumulate=acc, sample_offset=f * spp if acc else 0)
                ctx.render(cam, prm)
                ts.append(ctx.stats().kernel_ms)
            t = statistics.median(ts[1:])
            print(f"spp {spp:3d} accumulate {acc}: {t:7.3f} ms  {w * h * spp / t / 1e6:8.1f} Gsamples/s-ish ({w*h*spp/(t*1e-3)/1e9:.2f} G samples/s)")
