import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_rtiow_amd as V
w, h = 40, 24
sph, mat = V.make_three_sphere_scene(False)
cam = V.camera_from_ubo(V.ubo_from_image(w, h))
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for depth in (10, 50):
        for spp in (4,) * 6:
            a = ctx.render(cam, V.make_params(w, h, spp=spp, max_depth=depth, seed=1, kernel=1)); sa = ctx.stats()
            b = ctx.render(cam, V.make_params(w, h, spp=spp, max_depth=depth, seed=1, kernel=2)); sb = ctx.stats()
            d = (a != b).any(axis=2)
            ys, xs = np.nonzero(d)
            print(f"depth {depth} spp {spp}: segs v1 {sa.segments} v2 {sb.segments} paths {sa.paths}/{sb.paths} diff pixels {d.sum()}",
                  list(zip(ys[:4].tolist(), xs[:4].tolist())), a[d][:2].tolist(), b[d][:2].tolist())
