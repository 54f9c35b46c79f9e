"""A 3-D cloud of spheres (not flat along any axis) at growing sizes: frame time and tests per segment of the default kernel --
the same sweep as tools/grid_stats.py for scenes that do not stand on a plane.  usage: cloud_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_rtiow_amd as V
w, h = 1200, 800
cam = V.make_camera((0, 0, 9), (0, 0, 0), (0, 1, 0), 40.0, w / h, 0.05, 9.0)
for n in (100, 300, 560, 700, 1000, 1500, 2000, 3000, 4000, 5000, 6000):
    rng = np.random.default_rng(n)
    sph = np.zeros(n, V.SPHERE_DTYPE)
    mat = np.zeros(n, V.MATERIAL_DTYPE)
    pos = rng.uniform(-3, 3, (n, 3))
    sph["cx"], sph["cy"], sph["cz"] = pos[:, 0], pos[:, 1], pos[:, 2]
    sph["radius"] = rng.uniform(0.04, 0.08, n) * (300.0 / n) ** (1 / 3)
    kinds = rng.choice([0, 1, 2], n, p=[0.7, 0.2, 0.1])
    mat["kind"] = kinds
    mat["albedo"] = rng.uniform(0.3, 0.9, (n, 3))
    mat["ior"] = np.where(kinds == 2, 1.5, 0)
    with V.Context(0) as ctx:
        ctx.set_scene(sph, mat)
        prm = V.make_params(w, h, spp=32, max_depth=50, seed=1)
        for _ in range(4):
            ctx.render(cam, prm)
        st = ctx.stats()
        print(f"cloud of {n:5d} spheres, kernel {ctx.last_kernel()}: {st.kernel_ms:6.2f} ms, segments {st.segments}, tests/segment {st.sphere_tests / st.segments:.1f}", flush=True)
