"""Not collected by pytest; run by hand on a GPU box: the whole C4 frame (cover scene 1200x800, 500 spp,
depth 50) on every persistent kernel against the oracle, byte for byte (about 40 s of CPU on 16 cores)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vulkan_rtiow_amd as V
import oracle_bind
w, h, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 500
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
prm = dict(spp=spp, max_depth=50, seed=1, quantiser=V.RT_QUANT_BOOK)
t0 = time.time()
want, segs = oracle_bind.load().render(sph, mat, cam, V.make_params(w, h, **prm))
print(f"oracle: {time.time() - t0:.1f} s, {segs} segments", flush=True)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for kern in (2, 3):
        got = ctx.render(cam, V.make_params(w, h, kernel=kern, **prm))
        st = ctx.stats()
        print(f"kernel {kern}: {st.kernel_ms:.1f} ms, differing pixels {int((got != want).any(axis=2).sum())}, "
              f"segments equal {st.segments == segs}")
