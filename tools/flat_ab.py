"""Flat-axis box test on / off (RTIOW_DEBUG_FLAT=0 at launch: whole boxes) interleaved in one process: cover frame, 1/8
tile, 1 spp, C5 at 16 spp.  Frames must be identical.  usage: flat_ab.py [rounds]"""
import os, sys, statistics, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTIOW_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-rtiow_amd", "librtiow_hip_knobs.so"))  # the RTIOW_DEBUG_* knobs exist in this build only
import vulkan_rtiow_amd as V
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
cases = [("cover 1200x800x100", 11, 1200, 800, 100, 1), ("cover tile 0 of 8", 11, 1200, 800, 100, 8), ("cover 1200x800x1", 11, 1200, 800, 1, 1),
         ("C5 3840x2160x16", 32, 3840, 2160, 16, 1)]
for name, grid, w, h, spp, G in cases:
    sph, mat = V.make_cover_scene(1, grid)
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    ctxs = {}
    for v in ("flat", "whole boxes"):
        ctxs[v] = V.Context(0)
        ctxs[v].set_scene(sph, mat)
    prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=0, tile_count=G)
    t = {v: [] for v in ctxs}
    crc = {}
    for r in range(rounds + 2):
        for v in ctxs:
            os.environ["RTIOW_DEBUG_FLAT"] = "0" if v == "whole boxes" else "2"
            img = ctxs[v].render(cam, prm)
            st = ctxs[v].stats()
            if r >= 2:
                t[v].append(st.kernel_ms)
            crc[v] = (zlib.crc32(img.tobytes()), st.segments, st.sphere_tests)
    for v in ctxs:
        print(f"{name:22s} {v:12s} median {statistics.median(t[v]):9.3f} ms  min {min(t[v]):9.3f}  tests/segment {crc[v][2] / crc[v][1]:.1f}", flush=True)
        ctxs[v].close()
    assert crc["flat"][:2] == crc["whole boxes"][:2], crc
    print(f"{name:22s} frames identical", flush=True)
