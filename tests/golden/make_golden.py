"""Regenerates the committed golden fixtures.  Run in the build container only
(`python tests/golden/make_golden.py`): part (1) reads the reference's JPEG
screenshots under /root/reference, which do not exist on the GPU box.

 (1) ref_jpeg_stats.json   measurements of the reference's own result fixtures
                           (RTCHAP05/RTCHAP05/21986.jpg and 1728.jpg, RT01/RT01/4068.jpg): data derived
                           from images, no reference source text.
 (2) ch_known_answers.json the SURVEY.md section 8(c) known-answer table (derived in the survey
                           session from a separate non-fused float32 numpy restatement).
 (3) path_oracle_crc.json  CRC32 + segment counts of small PATH frames rendered by the
                           oracle itself (regression pins; PATH has no reference fixture).
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def jpeg_stats():
    from PIL import Image
    out = {}
    img = np.asarray(Image.open(os.path.join(REF, "RTCHAP05/RTCHAP05/21986.jpg")).convert("RGB")).astype(int)
    red = (img[..., 0] > 200) & (img[..., 1] < 60) & (img[..., 2] < 60)
    ys, xs = np.nonzero(red)
    h, w = img.shape[:2]
    out["RTCHAP05/RTCHAP05/21986.jpg"] = {
        "size": [w, h], "orientation": "display (row 0 = top of picture)",
        "red_rule": "r>200 and g<60 and b<60",
        "red_count": int(red.sum()),
        "red_bbox_x": [int(xs.min()), int(xs.max())], "red_bbox_y": [int(ys.min()), int(ys.max())],
        "top_left": img[0, 0].tolist(), "bottom_left": img[h - 1, 0].tolist(),
        "top_right": img[0, w - 1].tolist(), "bottom_right": img[h - 1, w - 1].tolist(),
        "jpeg_tolerance": 3,
    }
    # the older state of the same program: a 1024^2 compute image with viewportHeight = 2 shown in the 800x600
    # window (the quad of main.cpp:174-179 stretches it), so the sphere is an ellipse: pins the UBO path with a
    # non-square viewport-to-window mapping
    img = np.asarray(Image.open(os.path.join(REF, "RTCHAP05/RTCHAP05/1728.jpg")).convert("RGB")).astype(int)
    red = (img[..., 0] > 200) & (img[..., 1] < 60) & (img[..., 2] < 60)
    ys, xs = np.nonzero(red)
    h, w = img.shape[:2]
    out["RTCHAP05/RTCHAP05/1728.jpg"] = {
        "size": [w, h], "orientation": "display (row 0 = top of picture)", "compute_image": [1024, 1024],
        "ubo": [1024.0, 1024.0, 2.0, 2.0, 1.0], "red_rule": "r>200 and g<60 and b<60",
        "red_count": int(red.sum()),
        "red_bbox_x": [int(xs.min()), int(xs.max())], "red_bbox_y": [int(ys.min()), int(ys.max())],
        "top_left": img[0, 0].tolist(), "bottom_left": img[h - 1, 0].tolist(),
        "top_right": img[0, w - 1].tolist(), "bottom_right": img[h - 1, w - 1].tolist(),
        "centre": img[h // 2, w // 2].tolist(), "bbox_tolerance": 2, "count_tolerance": 0.005, "jpeg_tolerance": 3,
    }
    img = np.asarray(Image.open(os.path.join(REF, "RT01/RT01/4068.jpg")).convert("RGB")).astype(int)
    h, w = img.shape[:2]
    out["RT01/RT01/4068.jpg"] = {"size": [w, h], "top_mid": img[0, w // 2].tolist(),
                                 "bottom_mid": img[h - 1, w // 2].tolist(), "jpeg_tolerance": 3}
    return out


def known_answers():
    # image-space rows (row 0 = scene bottom), RGB bytes
    rows = [
        ("CH05", 800, 608, 167084, [169, 630], [73, 534], [221, 235, 255], [161, 199, 255], [255, 0, 0], [255, 0, 0], [255, 0, 0]),
        ("CH06", 800, 608, 167084, [169, 630], [73, 534], [221, 235, 255], [161, 199, 255], [128, 128, 255], [128, 30, 209], [30, 128, 209]),
        ("CH05", 400, 225, 41450, [85, 314], [0, 224], [215, 231, 255], [168, 203, 255], [255, 0, 0], [255, 0, 0], [255, 0, 0]),
        ("CH06", 400, 225, 41450, [85, 314], [0, 224], [215, 231, 255], [168, 203, 255], [128, 128, 255], [128, 41, 221], [37, 128, 217]),
    ]
    keys = ["mode", "width", "height", "hit_px", "bbox_x", "bbox_y", "px00", "pxWH", "centre",
            "mid_ymin5", "xmin5_mid"]
    return [dict(zip(keys, r)) for r in rows]


PATH_CASES = [
    # name, scene, width, height, spp, depth, seed, chunk, quantiser
    ("three_40x24_s4", "three", 40, 24, 4, 50, 1, 0, 1),
    ("three_bubble_33x19_s5_c2", "three_bubble", 33, 19, 5, 12, 7, 2, 0),
    ("cover_48x32_s2", "cover11", 48, 32, 2, 50, 1, 0, 1),
    ("cover_lens_30x20_s3_c1", "cover3", 30, 20, 3, 8, 3, 1, 1),
]


def build_case(V, orc, scene, w, h):
    if scene == "three":
        sph, mat = orc.make_three_sphere_scene(False)
        cam = orc.camera_from_ubo(orc.ubo_from_image(w, h))
    elif scene == "three_bubble":
        sph, mat = orc.make_three_sphere_scene(True)
        cam = orc.camera_from_ubo(orc.ubo_from_image(w, h))
    else:
        sph, mat = orc.make_cover_scene(1, int(scene[5:]))
        cam = orc.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    return sph, mat, cam


def path_crcs():
    import oracle_bind
    import vulkan_rtiow_amd as V
    orc = oracle_bind.load()
    out = {}
    for name, scene, w, h, spp, depth, seed, chunk, quant in PATH_CASES:
        sph, mat, cam = build_case(V, orc, scene, w, h)
        prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=seed, chunk_spp=chunk, quantiser=quant)
        img, segs = orc.render(sph, mat, cam, prm)
        out[name] = {"scene": scene, "width": w, "height": h, "spp": spp, "max_depth": depth,
                     "seed": seed, "chunk_spp": chunk, "quantiser": quant, "n_spheres": int(len(sph)),
                     "scene_crc32": zlib.crc32(sph.tobytes() + mat.tobytes()),
                     "frame_crc32": zlib.crc32(img.tobytes()), "segments": int(segs)}
    return out


if __name__ == "__main__":
    if os.path.isdir(REF):
        json.dump(jpeg_stats(), open(os.path.join(HERE, "ref_jpeg_stats.json"), "w"), indent=1)
    json.dump(known_answers(), open(os.path.join(HERE, "ch_known_answers.json"), "w"), indent=1)
    json.dump(path_crcs(), open(os.path.join(HERE, "path_oracle_crc.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)
