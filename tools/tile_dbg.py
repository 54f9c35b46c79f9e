import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V
w, h, spp = 1200, 800, 100
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for G in (1, 8):
        prm = V.make_params(w, h, spp=spp, max_depth=int(os.environ.get("DEPTH", "50")), seed=1, row_block=4, tile_rank=0, tile_count=G)
        for _ in range(2):
            ctx.render(cam, prm)
        st = ctx.stats(); d = list(st.debug)
        sparse = d[2] >> 32; d[2] &= 0xFFFFFFFF
        iters = max(1, d[2])
        tot = max(1, d[3] + d[4] + d[6])
        print(f"G={G}: {st.kernel_ms:.2f} ms segs {st.segments} wave-iters {d[2]} (sparse {sparse}) util {st.segments/(iters*128):.3f} "
              f"cycles/iter/wave {tot/iters:.0f} clock {d[7]} MHz shares refill {d[3]/tot:.3f} fast {(d[4]-d[5])/tot:.3f} slow {d[5]/tot:.3f} shade {d[6]/tot:.3f}")
