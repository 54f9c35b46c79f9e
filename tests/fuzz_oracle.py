"""Not collected by pytest; run by hand on a GPU box: python tests/fuzz_oracle.py [cases] [first_seed]
One-off fuzz (GPU + CPU oracle): the cases of tools/fuzz_kernels.py with at most 900 spheres, every GPU
kernel against the oracle, byte for byte.  usage: fuzz_oracle.py [cases] [first_seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # (this file lives in tests/: it uses the oracle)
import vulkan_rtiow_amd as V
from test_gpu_random_scenes import fuzz_case
import oracle_bind

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 700000
orc = oracle_bind.load()
bad = done = 0
t0 = time.time()
with V.Context(0) as ctx:
    for case in range(first, first + cases):
        sph, mat, cam, w, h, base = fuzz_case(case)
        if len(sph) > 900:
            continue
        want, segs = orc.render(sph, mat, cam, V.make_params(w, h, **base))
        ctx.set_scene(sph, mat)
        for kern in (1, 2, 3, 4):
            got = ctx.render(cam, V.make_params(w, h, kernel=kern, **base))
            if int((got != want).any(axis=2).sum()) or ctx.stats().segments != segs:
                bad += 1
                print(f"MISMATCH case {case} kernel {kern} n={len(sph)} {w}x{h}", flush=True)
        done += 1
        if done % 200 == 0:  # (a run that writes nothing for minutes is taken to be hung)
            print(f"... {done} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"{done} cases x 4 kernels (4 = the clustered list with the primary pass forced on) against the oracle, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
