"""Re-runs one case of tools/fuzz_kernels.py against the oracle and prints where the kernels differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # (this file lives in tests/: it uses the oracle)
import vulkan_rtiow_amd as V
from test_gpu_random_scenes import fuzz_case
import oracle_bind

case = int(sys.argv[1])
sph, mat, cam, w, h, base = fuzz_case(case)
n = len(sph)
print('n', n, 'size', w, h, base)
orc = oracle_bind.load()
want, segs = orc.render(sph, mat, cam, V.make_params(w, h, **base))
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    for rep in range(3):
        for kname, kern in (("pixel", 1), ("flat", 2), ("clustered", 3)):
            got = ctx.render(cam, V.make_params(w, h, kernel=kern, **base))
            st = ctx.stats()
            d = np.argwhere((got != want).any(axis=2))
            print(rep, kname, "last_kernel", ctx.last_kernel(), "segments", st.segments, "oracle", segs, "diff pixels", len(d),
                  [(int(y), int(x), got[y, x].tolist(), want[y, x].tolist()) for y, x in d[:3]])

if len(sys.argv) > 2:   # depth sweep at one pixel: first bounce limit at which flat and clustered differ
    py, px = int(sys.argv[2]), int(sys.argv[3])
    with V.Context(0) as ctx:
        ctx.set_scene(sph, mat)
        for d in range(1, 20):
            b = dict(base); b["max_depth"] = d
            f = ctx.render(cam, V.make_params(w, h, kernel=2, **b)); sf = ctx.stats().segments
            c = ctx.render(cam, V.make_params(w, h, kernel=3, **b)); sc = ctx.stats().segments
            print("depth", d, "flat", f[py, px].tolist(), "clustered", c[py, px].tolist(), "segments", sf, sc,
                  "differing pixels", int((f != c).any(axis=2).sum()))
