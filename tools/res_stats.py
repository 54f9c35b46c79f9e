"""The cover frame at other resolutions and sample counts: ms per megapixel-sample should not depend on them.
usage: res_stats.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
sph, mat = V.make_cover_scene(1, 11)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for w, h, spp in ((320, 180, 100), (640, 360, 100), (1000, 667, 100), (1200, 800, 100), (1280, 720, 100), (1920, 1080, 100), (2560, 1440, 100),
                      (3840, 2160, 100), (1200, 800, 30), (1200, 800, 64), (1200, 800, 99), (1200, 800, 128), (1200, 800, 250), (1201, 799, 100), (1234, 777, 77)):
        cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1)
        ts = []
        for _ in range(5):
            ctx.render(cam, prm)
            ts.append(ctx.stats().kernel_ms)
        ms = statistics.median(ts[2:])
        st = ctx.stats()
        print(f"{w:5d} x {h:4d} x {spp:4d} spp: {ms:8.3f} ms  {ms / (w * h * spp / 1e8):6.3f} ms per 1e8 samples  {st.segments / ms / 1e6:6.2f} G segments/s  tests/segment {st.sphere_tests / st.segments:.1f}", flush=True)
