import os, sys, statistics, zlib
sys.path.insert(0, "/root/repo")
import vulkan_rtiow_amd as V
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
for name, spp, G, rank in (("tile8 r0", 100, 8, 0), ("tile8 r5", 100, 8, 5), ("full", 100, 1, 0), ("1spp", 1, 1, 0), ("4spp", 4, 1, 0), ("tile4 r0", 100, 4, 0)):
    ctxs = {v: V.Context(0) for v in ("pairs", "no pairs")}
    for c in ctxs.values():
        c.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=rank, tile_count=G)
    t = {v: [] for v in ctxs}; crc = {}
    for r in range(11):
        for v, c in ctxs.items():
            if v == "no pairs": os.environ["RTIOW_DEBUG_NO_TAIL_PAIRS"] = "1"
            else: os.environ.pop("RTIOW_DEBUG_NO_TAIL_PAIRS", None)
            img = c.render(cam, prm); st = c.stats()
            if r >= 2: t[v].append(st.kernel_ms)
            crc[v] = (zlib.crc32(img.tobytes()), st.segments)
    assert crc["pairs"] == crc["no pairs"], crc
    print(name, " ".join(f"{v}: median {statistics.median(t[v]):.3f} min {min(t[v]):.3f}" for v in ctxs), flush=True)
    for c in ctxs.values(): c.close()
