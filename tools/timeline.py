"""Per-wave timeline of the persistent kernel from a -DRTIOW_DEBUG_TIMELINE build (three wall-clock stamps per
wave: queue dry, first sparse iteration, end; 50 us bins).  usage:
    RTIOW_DEBUG_HIST=1 RTIOW_LIB=$PWD/vulkan-rtiow_amd/librtiow_hip_tl.so python tools/timeline.py [G ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
w, h, spp = 1200, 800, 100
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for G in [int(x) for x in sys.argv[1:]] or [8]:
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=0, tile_count=G)
        for _ in range(3):
            ctx.render(cam, prm)
        sys.stderr.write(f"--- G={G} (third frame; the chunk order is the second frame's)\n")
        st = ctx.stats()
        print(f"G={G}: {st.kernel_ms:.3f} ms segs {st.segments}")
