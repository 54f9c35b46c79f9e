"""One-off fuzz (GPU) of the reference's shaders: random UBOs and image sizes through the default row kernel (two-phase pixels where
the UBO allows them) against the row kernel with hipcc's full square roots and quotients (RTIOW_DEBUG_CH_FULL, knobs build), byte for
byte, both shaders.  The full-form kernel is pinned to the oracle by tests/; this widens the UBO / size space cheaply.
usage: fuzz_ch.py [cases] [first_seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_rtiow_amd as V

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
edges = [2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 259, 511, 513, 1023, 1024, 1025, 1027]
bad = pixels = 0
t0 = time.time()
with V.Context(0, lib_path=V.api.KNOBS_LIB_PATH) as ctx:
    for case in range(first, first + cases):
        rng = np.random.default_rng(case)
        kind = case % 8
        w = int(rng.choice(edges)) if rng.random() < 0.3 else int(rng.integers(2, 3000))
        h = int(rng.choice(edges)) if rng.random() < 0.3 else int(rng.integers(2, 2000))
        u = V.ubo_from_image(w, h)
        if kind >= 2:  # anywhere in the lean range, log-uniform; kinds 6, 7: the sphere about the size of the frame
            span = 4 if kind >= 6 else 19
            u.viewportWidth = float(np.float32(2.0 ** rng.uniform(-span, span)))
            u.viewportHeight = float(np.float32(2.0 ** rng.uniform(-span, span)))
            u.focalLength = float(np.float32(2.0 ** rng.uniform(-span, span)))
            if rng.random() < 0.1:
                u.focalLength = -u.focalLength  # the sphere behind the camera
            if rng.random() < 0.1:
                u.viewportHeight = -u.viewportHeight
        mode = V.RT_MODE_CH06 if case % 3 else V.RT_MODE_CH05
        os.environ.pop("RTIOW_DEBUG_CH_FULL", None)
        got = ctx.render_ubo(u, mode)
        os.environ["RTIOW_DEBUG_CH_FULL"] = "1"
        want = ctx.render_ubo(u, mode)
        os.environ.pop("RTIOW_DEBUG_CH_FULL", None)
        pixels += w * h
        diff = int((got != want).any(axis=2).sum())
        if diff:
            bad += 1
            print(f"MISMATCH case {case}: {w}x{h} mode {mode} ubo {u.viewportWidth} {u.viewportHeight} {u.focalLength}: {diff} pixels")
print(f"{cases} cases from seed {first}, {pixels / 1e6:.0f} Mpixel, {bad} mismatches, {time.time() - t0:.0f} s")
