import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
w, h = 1200, 800
for grid in [int(x) for x in os.environ.get("GRIDS", "11,14,16,20,22,24,28,30,32,36").split(",")]:
    sph, mat = V.make_cover_scene(1, grid)
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    with V.Context(0) as ctx:
        ctx.set_scene(sph, mat)
        prm = V.make_params(w, h, spp=32, max_depth=50, seed=1)
        for _ in range(4):
            ctx.render(cam, prm)
        st = ctx.stats()
        print(f"grid {grid}: {len(sph)} spheres kernel {ctx.last_kernel()}: {st.kernel_ms:.2f} ms, segments {st.segments}, tests/segment {st.sphere_tests / st.segments:.1f}", flush=True)
