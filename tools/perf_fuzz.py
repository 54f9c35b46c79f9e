"""Random scenes of several layouts and sizes through the default kernel and the flat list: time, tests per segment and the ratio of
the two kernels -- a search for layouts the clustered list handles badly (as the super-clusters of scenes whose cluster count is no
power of two were: profiles/r03_scene_size_sweep.txt).  usage: perf_fuzz.py [cases]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_rtiow_amd as V
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w, h = 600, 400
rows = []
with V.Context(0) as ctx:
    for case in range(cases):
        rng = np.random.default_rng(1000 + case)
        n = int(10 ** rng.uniform(2.0, 3.7))
        layout = ["plane_y", "plane_x", "plane_z", "cloud", "line", "two_blobs", "shell", "thick_slab", "mixed_radii"][case % 9]
        pos = rng.uniform(-1, 1, (n, 3))
        rad = np.full(n, 0.5 / n ** 0.5)
        if layout.startswith("plane"):
            ax = "xyz".index(layout[-1]); pos[:, ax] *= 0.002
        elif layout == "cloud":
            rad = np.full(n, 0.35 / n ** (1 / 3))
        elif layout == "line":
            pos[:, 1] *= 0.01; pos[:, 2] *= 0.01; rad = np.full(n, 0.5 / n)
        elif layout == "two_blobs":
            pos *= 0.3; pos[: n // 2, 0] -= 0.7; pos[n // 2:, 0] += 0.7; rad = np.full(n, 0.12 / n ** (1 / 3))
        elif layout == "shell":
            pos /= np.linalg.norm(pos, axis=1, keepdims=True); rad = np.full(n, 1.2 / n ** 0.5)
        elif layout == "thick_slab":
            pos[:, 1] *= 0.15; rad = np.full(n, 0.4 / n ** 0.5)
        elif layout == "mixed_radii":
            pos[:, 1] *= 0.002; rad = rad * rng.choice([0.3, 1.0, 2.5], n)
        sph = np.zeros(n + 1, V.SPHERE_DTYPE); mat = np.zeros(n + 1, V.MATERIAL_DTYPE)
        sph["cx"][1:], sph["cy"][1:], sph["cz"][1:] = pos[:, 0], pos[:, 1], pos[:, 2]
        sph["radius"][1:] = rad
        kinds = rng.choice([0, 1, 2], n, p=[0.7, 0.2, 0.1])
        mat["kind"][1:] = kinds; mat["albedo"][1:] = rng.uniform(0.3, 0.9, (n, 3)); mat["ior"][1:] = np.where(kinds == 2, 1.5, 0)
        sph[0] = (0.0, -1001.2, 0.0, 1000.0); mat[0] = (0, (0.5, 0.5, 0.5), 0.0, 0.0, (0, 0))
        cam = V.make_camera((2.6, 1.2, 2.2), (0, 0, 0), (0, 1, 0), 35.0, w / h, 0.02, 3.5)
        ctx.set_scene(sph, mat)
        out = {}
        for k in (0, 2):
            prm = V.make_params(w, h, spp=32, max_depth=50, seed=1, kernel=k)
            ts = []
            for _ in range(4):
                ctx.render(cam, prm); ts.append(ctx.stats().kernel_ms)
            st = ctx.stats()
            out[k] = (statistics.median(ts[1:]), st.sphere_tests / st.segments, st.segments, ctx.last_kernel())
        rows.append((layout, n, out))
        print(f"{case:3d} {layout:12s} n={n:5d}: default (kernel {out[0][3]}) {out[0][0]:7.3f} ms  {out[0][1]:6.1f} tests/seg | flat {out[2][0]:7.3f} ms | flat/default {out[2][0] / out[0][0]:5.2f}  "
              f"segs {out[0][2]}  ns per segment {out[0][0] * 1e6 / out[0][2]:.3f}", flush=True)
