"""One-off fuzz (GPU): random scenes through the clustered kernel against the flat one, byte for byte.
The flat kernel is pinned to the oracle by tests/; this widens the scene/camera space cheaply.
usage: fuzz_kernels.py [cases] [first_seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import vulkan_rtiow_amd as V
from test_gpu_random_scenes import fuzz_case

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = 0
t0 = time.time()
with V.Context(0) as ctx:
    for case in range(first, first + cases):
        sph, mat, cam, w, h, base = fuzz_case(case)
        n = len(sph)
        ctx.set_scene(sph, mat)
        flat = ctx.render(cam, V.make_params(w, h, kernel=V.KERNEL_PERSISTENT, **base))
        sf = ctx.stats()
        clus = ctx.render(cam, V.make_params(w, h, kernel=V.KERNEL_CLUSTERED, **base))
        sc = ctx.stats()
        k = ctx.last_kernel()
        diff = int((flat != clus).any(axis=2).sum())
        if diff or sf.segments != sc.segments:
            bad += 1
            print(f"MISMATCH case {case}: n={n} {w}x{h} pixels={diff} segs {sf.segments} vs {sc.segments}")
        # the clustered list with the primary pass forced on (at these sample counts the default leaves it off)
        pas = ctx.render(cam, V.make_params(w, h, kernel=V.KERNEL_CLUSTERED_PASS, **base))
        sp = ctx.stats()
        diff = int((flat != pas).any(axis=2).sum())
        if diff or sf.segments != sp.segments:
            bad += 1
            print(f"MISMATCH (primary pass) case {case}: n={n} {w}x{h} pixels={diff} segs {sf.segments} vs {sp.segments}")
        if case % 3 == 0:   # the same frame as row tiles and as two accumulated dispatches, default kernel choice
            rng = np.random.default_rng(case + 7)
            count, block = int(rng.choice([2, 3, 5])), int(rng.choice([1, 3, 4, 16]))
            tiled = np.zeros_like(flat)
            for r in range(count):
                nrows = V.tile_row_count(h, block, r, count)
                rows = [V.tile_global_row(k, block, r, count) for k in range(nrows)]
                if len(rows) == 0:
                    continue
                part = ctx.render(cam, V.make_params(w, h, row_block=block, tile_rank=r, tile_count=count,
                                                     kernel=V.KERNEL_CLUSTERED_PASS if case % 2 else 0, **base))
                tiled[rows] = part
            if (tiled != flat).any():
                bad += 1
                print(f"MISMATCH (tiles {count}x{block}) case {case}")
            if base["spp"] >= 2:
                a_spp = base["spp"] // 2
                b1 = dict(base); b1["spp"] = a_spp
                b2 = dict(base); b2["spp"] = base["spp"] - a_spp
                kk = V.KERNEL_CLUSTERED_PASS if case % 2 else 0
                ctx.render(cam, V.make_params(w, h, accumulate=1, sample_offset=0, kernel=kk, **b1))
                acc = ctx.render(cam, V.make_params(w, h, accumulate=1, sample_offset=a_spp, kernel=kk, **b2))
                if (acc != flat).any():
                    bad += 1
                    print(f"MISMATCH (accumulate {a_spp}+{base['spp'] - a_spp}) case {case}")
        if (case - first) % 25 == 0:
            print(f"case {case}: n={n} kernel={k} tests/seg flat {sf.sphere_tests / max(1, sf.segments):.0f} "
                  f"clustered {sc.sphere_tests / max(1, sc.segments):.1f} with the primary pass {sp.sphere_tests / max(1, sp.segments):.1f}  ({time.time() - t0:.0f} s)", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
