"""Kernel time of every BASELINE configuration and of the small-work cases on one GPU, one line each -- the table of
DESIGN 5.1 (run after a kernel change: a pool-size change once tripled C2 unnoticed).  usage: all_configs.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V


def run(ctx, name, cam, frames=5, **kw):
    prm = V.make_params(**kw)
    ts = []
    for _ in range(frames):
        ctx.render(cam, prm)
        ts.append(ctx.stats().kernel_ms)
    st = ctx.stats()
    print(f"{name:58s} {statistics.median(ts[2:]):9.3f} ms  kernel {ctx.last_kernel()}  tests/segment "
          f"{st.sphere_tests / max(1, st.segments):6.1f}", flush=True)


with V.Context(0) as ctx:
    sph, mat = V.make_three_sphere_scene(False)
    ctx.set_scene(sph, mat)
    run(ctx, "C2 three spheres 400x225x100spp", V.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 400 / 225, 0.0, 1.0),
        width=400, height=225, spp=100, max_depth=50, seed=1)
    sph, mat = V.make_cover_scene(1, 11)
    ctx.set_scene(sph, mat)
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    for spp in (1, 4, 16, 100, 500):
        run(ctx, f"C3/C4 cover 1200x800x{spp}spp", cam, width=1200, height=800, spp=spp, max_depth=50, seed=1)
    run(ctx, "C3 cover, flat list (kernel 2)", cam, frames=4, width=1200, height=800, spp=100, max_depth=50, seed=1, kernel=2)
    for G in (2, 4, 8):
        run(ctx, f"C3 cover, tile 0 of {G} (row blocks of 4)", cam, width=1200, height=800, spp=100, max_depth=50, seed=1,
            row_block=4, tile_rank=0, tile_count=G)
    run(ctx, "cover 300x200x10spp", V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0),
        width=300, height=200, spp=10, max_depth=50, seed=1)
    sph, mat = V.make_cover_scene(1, 32)
    ctx.set_scene(sph, mat)
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 3840 / 2160, 0.1, 10.0)
    run(ctx, "C5 4099 spheres 3840x2160x16spp", cam, frames=4, width=3840, height=2160, spp=16, max_depth=50, seed=1)
    run(ctx, "C5 4099 spheres 3840x2160x1024spp (whole frame)", cam, frames=3, width=3840, height=2160, spp=1024, max_depth=50, seed=1)
