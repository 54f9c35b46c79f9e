"""N cover frames (1200x800x100 spp, depth 50) back to back on one context; the frame's CRC-32 every 100th frame against the first.
usage: python tools/soak.py [frames]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_rtiow_amd as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
w, h, spp = 1200, 800, 100
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1)
t0 = time.time()
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    ms, bad, first = [], 0, None
    for f in range(n):
        img = ctx.render(cam, prm)
        ms.append(ctx.stats().kernel_ms)
        if f % 100 == 0:
            crc = zlib.crc32(np.ascontiguousarray(img).tobytes())
            first = crc if first is None else first
            bad += crc != first
ms = np.array(ms)
print(f"{n} cover frames ({w}x{h}x{spp} spp, depth 50) back to back on one context; CRC-32 of the frame checked every 100th frame against the first; RtStats.kernel_ms:")
print(f"frames {n}  median {np.median(ms):.3f} ms  min {ms.min():.3f}  max {ms.max():.3f} (the first frames: no chunk order yet)  crc mismatches {bad}   ({time.time() - t0:.0f} s)")
print("medians of the last hundred frames at every 500th frame: " + " ".join(f"{np.median(ms[k - 100:k]):.3f}" for k in range(500, n + 1, 500)))
