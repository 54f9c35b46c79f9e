"""Express lane on / off (RTIOW_DEBUG_EXPRESS, read at every launch) on the cover frame, interleaved in one process:
whole frame, 1/G tiles, low-spp frames.  Frames must be identical; prints kernel ms (median of the warm launches).
usage: express_ab.py [--us 450] [--from 12] [--configs full,tile8,tile4,tile2,spp1,spp16]"""
import os, sys, statistics, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V

def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default

configs = arg("--configs", "tile8,full,tile4,tile2,spp1,spp16").split(",")
variants = [("off", {"RTIOW_DEBUG_EXPRESS": "0"})]
if "--variant-only" in sys.argv:
    variants.append(("express kernel, no x wave", {"RTIOW_DEBUG_EXPRESS": "2"}))
for us in arg("--us", "450").split(","):
    for frm in arg("--from", "12").split(","):
        variants.append((f"on us={us} from={frm}", {"RTIOW_DEBUG_EXPRESS": "1", "RTIOW_DEBUG_EXPRESS_US": us, "RTIOW_DEBUG_EXPRESS_FROM": frm}))
rounds = int(arg("--rounds", "7"))
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
shapes = {"full": (100, 1, 0), "tile8": (100, 8, 0), "tile8r5": (100, 8, 5), "tile4": (100, 4, 0), "tile2": (100, 2, 0), "spp1": (1, 1, 0), "spp16": (16, 1, 0),
          "spp4": (4, 1, 0)}
ctxs = {}
for name, _ in variants:  # one context per variant: each keeps its own chunk order
    ctxs[name] = V.Context(0)
    ctxs[name].set_scene(sph, mat)
for cfg in configs:
    spp, G, rank = shapes[cfg]
    prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=rank, tile_count=G)
    times = {n: [] for n, _ in variants}
    crc, segs = {}, {}
    for r in range(rounds + 2):
        for name, env in variants:
            os.environ.update(env)
            img = ctxs[name].render(cam, prm)
            st = ctxs[name].stats()
            if r >= 2:
                times[name].append(st.kernel_ms)
            crc[name] = zlib.crc32(img.tobytes())
            segs[name] = st.segments
    ok = len(set(crc.values())) == 1 and len(set(segs.values())) == 1
    for name, _ in variants:
        t = times[name]
        print(f"{cfg:8s} {name:24s} median {statistics.median(t):7.3f} ms  min {min(t):7.3f}  max {max(t):7.3f}", flush=True)
    print(f"{cfg:8s} frames {'identical' if ok else 'DIFFER ' + str(crc) + str(segs)}", flush=True)
    assert ok
