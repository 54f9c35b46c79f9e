"""Prints the kernel-internal counters of a diagnostic build (RTIOW_LIB=.../librtiow_hip_dbg.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
h = int(sys.argv[3]) if len(sys.argv) > 3 else 800
kernel = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sph, mat = V.make_cover_scene(1, int(os.environ.get("GRID", "11")))
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for chunk in (8,):
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, chunk_spp=chunk, kernel=kernel)
        ctx.render(cam, prm)
        st = ctx.stats()
        d = list(st.debug)
        sparse = d[2] >> 32
        d[2] &= 0xFFFFFFFF
        iters = max(1, d[2])
        print(f"take {chunk}: {st.kernel_ms:.2f} ms segs {st.segments} wave-iters {d[2]} "
              f"segments per iteration and slot {st.segments / (iters * 128):.3f} (all; the passes trace the camera segments: {(st.segments - st.paths) / (iters * 128):.3f} without -- kernel 3 at spp >= 8) "
              f"slow trips/iter {d[0] / iters:.2f} (per slot {d[0] / iters / 2:.2f}) "
              f"cands/segment {d[1] / max(1, st.segments):.2f} sparse iters {sparse} tests/segment {st.sphere_tests / max(1, st.segments):.1f}")
        tot = max(1, d[3] + d[4] + d[6])
        print(f"   cycle shares: refill {d[3] / tot:.3f} trace-fast {(d[4] - d[5]) / tot:.3f} trace-slow {d[5] / tot:.3f} "
              f"shade {d[6] / tot:.3f}; cycles/iter/wave {tot / iters:.0f}; shader clock ~{d[7]} MHz")
