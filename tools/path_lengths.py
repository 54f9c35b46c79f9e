"""Path-length distribution of the cover frame: segments(max_depth = d) - segments(d - 1) = paths that take a d-th
segment.  Prints the share of paths reaching depth d and the share of all segments taken beyond d."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
w, h, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 20
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    segs = {}
    for d in (1, 2, 3, 4, 6, 8, 10, 12, 13, 16, 20, 25, 30, 40, 49, 50):
        ctx.render(cam, V.make_params(w, h, spp=spp, max_depth=d, seed=1))
        segs[d] = ctx.stats().segments
    n = w * h * spp
    total = segs[50]
    print(f"paths {n}, segments at depth 50: {total} ({total / n:.3f} per path)")
    for d in sorted(segs):
        print(f"segments within depth {d:2d}: {segs[d] / total * 100:6.2f} %   beyond: {(total - segs[d]) / total * 100:6.2f} %  ({(total - segs[d]) / n:.4f} per path)")
    print(f"paths that take segment 50: {(segs[50] - segs[49]) / n * 100:.4f} %")
