"""Generates tests/golden/frame_golden.json: CRC-32 and path-segment count of the ORACLE's frame for every
bench.py workload, so that bench.py (N = 1, outside the timed region) and tests/test_gpu_fullsize.py can tie the
timed GPU frame to the oracle without running the oracle on the GPU box.

Run in the build container (minutes of CPU; the GPU box never runs it):

    python tests/golden/make_frame_golden.py [workload ...]

Whole frames for the workloads the CPU finishes in minutes; for cover4096_3840x2160_1024spp (about 10^14 sphere
tests) the four rows tests/test_gpu_fullsize.py::test_config5_* compares -- sky, horizon, field, foreground.
Parameters are bench.py's own (WORKLOADS, build_scene, seed 1, book quantiser): the script imports them.
The frame is the packed RGBA8 array [rows, width, 4] uint8, row 0 = scene bottom; crc32 = zlib.crc32 of its bytes.
"""
import json
import os
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import bench  # noqa: E402  (WORKLOADS / build_scene only: nothing of it touches a GPU at import)
import oracle_bind  # noqa: E402
import vulkan_rtiow_amd as V  # noqa: E402

OUT = os.path.join(HERE, "frame_golden.json")
ROW_SAMPLES = {"cover4096_3840x2160_1024spp": [270, 810, 1350, 1890]}  # (global rows, as test_config5_* uses)


def golden_for(orc, name):
    scene, grid_half, w, h, spp, depth = bench.WORKLOADS[name]
    sph, mat, cam = bench.build_scene(V, scene, grid_half, w, h)
    t0 = time.perf_counter()
    entry = {"width": w, "height": h, "spp": spp, "max_depth": depth, "seed": 1, "quantiser": "book",
             "spheres": int(len(sph)), "scene_crc32": zlib.crc32(np.ascontiguousarray(sph).tobytes())}
    if name in ROW_SAMPLES:
        rows = {}
        for row in ROW_SAMPLES[name]:
            # one row = tile `row` of an h-way split with blocks of one row
            prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, quantiser=V.RT_QUANT_BOOK, row_block=1,
                                tile_rank=row, tile_count=h)
            img, segs = orc.render(sph, mat, cam, prm)
            assert img.shape[0] == 1
            rows[str(row)] = {"crc32": zlib.crc32(img.tobytes()), "segments": int(segs)}
        entry["rows"] = rows
    else:
        prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, quantiser=V.RT_QUANT_BOOK)
        img, segs = orc.render(sph, mat, cam, prm)
        entry["crc32"] = zlib.crc32(img.tobytes())
        entry["segments"] = int(segs)
        entry["paths"] = w * h * spp
    entry["oracle_seconds"] = round(time.perf_counter() - t0, 1)
    return entry


def main():
    orc = oracle_bind.load()
    names = sys.argv[1:] or sorted(bench.WORKLOADS)
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {}
    doc["_about"] = ("oracle/rtiow_oracle.c frames of bench.py's workloads: crc32 = zlib.crc32 of the packed RGBA8 bytes "
                     "(rows bottom-up), segments = path segments traced; made by tests/golden/make_frame_golden.py")
    for name in names:
        doc[name] = golden_for(orc, name)
        print(name, doc[name], flush=True)
        json.dump(doc, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
