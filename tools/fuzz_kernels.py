"""One-off fuzz (GPU): random scenes through the clustered kernel against the flat one, byte for byte.
The flat kernel is pinned to the oracle by tests/; this widens the scene/camera space cheaply.
usage: fuzz_kernels.py [cases] [first_seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import vulkan_rtiow_amd as V
from test_gpu_random_scenes import _random_scene

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = 0
t0 = time.time()
with V.Context(0) as ctx:
    for case in range(first, first + cases):
        rng = np.random.default_rng(case)
        n = int(rng.choice([3, 17, 64, 65, 130, 400, 900, 1600, 2500, 4000, 6000]))
        scale = float(rng.choice([0.01, 1.0, 1.0, 50.0, 3000.0]))
        sph, mat = _random_scene(rng, n, scale)
        if rng.random() < 0.3:   # a few more very large spheres
            k = min(int(rng.integers(1, 6)), n - 1)
            if k > 0:
                sph["radius"][1:1 + k] = rng.uniform(5, 60, k) * scale
        w, h = int(rng.integers(16, 160)), int(rng.integers(9, 100))
        dist = float(rng.choice([0.3, 1.0, 2.0, 4.0, 8.0, 40.0])) * scale
        frm = rng.normal(size=3)
        frm = frm / np.linalg.norm(frm) * dist + np.array([0, 0.5 * scale * rng.random(), 0])
        cam = V.make_camera(tuple(frm), (0.0, 0.0, 0.0), (0, 1, 0), float(rng.uniform(15, 110)), w / h,
                            float(rng.choice([0.0, 0.02, 0.3])) * scale, max(dist, 1e-3))
        base = dict(spp=int(rng.integers(1, 9)), max_depth=int(rng.choice([2, 8, 50])), seed=int(rng.integers(0, 2**31)),
                    quantiser=int(rng.integers(0, 2)))
        ctx.set_scene(sph, mat)
        flat = ctx.render(cam, V.make_params(w, h, kernel=V.KERNEL_PERSISTENT, **base))
        sf = ctx.stats()
        clus = ctx.render(cam, V.make_params(w, h, kernel=V.KERNEL_CLUSTERED, **base))
        sc = ctx.stats()
        k = ctx.last_kernel()
        diff = int((flat != clus).any(axis=2).sum())
        if diff or sf.segments != sc.segments:
            bad += 1
            print(f"MISMATCH case {case}: n={n} scale={scale} {w}x{h} dist={dist} pixels={diff} segs {sf.segments} vs {sc.segments}")
        if (case - first) % 25 == 0:
            print(f"case {case}: n={n} kernel={k} tests/seg flat {sf.sphere_tests / max(1, sf.segments):.0f} "
                  f"clustered {sc.sphere_tests / max(1, sc.segments):.1f}  ({time.time() - t0:.0f} s)", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
