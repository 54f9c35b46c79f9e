"""GPU (-m gpu): the C++ host harness (the stand-in for RTCHAP06/main.cpp's loop) end to end."""
import os
import subprocess
import sys

import numpy as np
import pytest

import vulkan_rtiow_amd as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "vulkan-rtiow_amd", "rtiow_main")


def _read_ppm(path):
    data = open(path, "rb").read()
    magic, dims, maxv, body = data.split(b"\n", 3)
    w, h = map(int, dims.split())
    assert magic == b"P6" and maxv == b"255"
    return np.frombuffer(body, np.uint8).reshape(h, w, 3)


def test_harness_ch06_matches_oracle(oracle, tmp_path):
    out = str(tmp_path / "ch06.ppm")
    subprocess.run([MAIN, "--scene", "ch06", "--width", "400", "--height", "225", "--out", out], check=True)
    want = oracle.render_ubo(oracle.ubo_from_image(400, 225), V.RT_MODE_CH06)
    assert np.array_equal(_read_ppm(out), want[::-1, :, :3])   # the writer flips like rt.frag:8


def test_harness_png_output(oracle, tmp_path):
    """--out x.png: the same picture through rtWritePNG (stored deflate), decoded here with zlib."""
    import zlib
    out = str(tmp_path / "ch05.png")
    subprocess.run([MAIN, "--scene", "ch05", "--width", "160", "--height", "90", "--out", out], check=True)
    data = open(out, "rb").read()
    idat = data[data.index(b"IDAT") + 4:data.index(b"IEND") - 8]   # chunk body (CRC and next length cut off)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(90, 1 + 3 * 160)
    want = oracle.render_ubo(oracle.ubo_from_image(160, 90), V.RT_MODE_CH05)
    assert np.array_equal(raw[:, 1:].reshape(90, 160, 3), want[::-1, :, :3])


def test_harness_scene_file_progressive_matches_oracle(oracle, tmp_path):
    """scene file + camera line, 3 progressive frames of 4 spp == the oracle at 12 spp."""
    out = str(tmp_path / "demo.ppm")
    scene = os.path.join(ROOT, "tests", "golden", "demo_scene.txt")
    w, h = 96, 64
    subprocess.run([MAIN, "--scene", "file", "--file", scene, "--width", str(w), "--height", str(h), "--spp", "4",
                    "--frames", "3", "--progressive", "1", "--depth", "50", "--seed", "1", "--out", out], check=True)
    sph = np.zeros(5, V.SPHERE_DTYPE)
    mat = np.zeros(5, V.MATERIAL_DTYPE)
    rows = [(0, -100.5, -1, 100, 0, (0.8, 0.8, 0.0), 0, 0), (0, 0, -1, 0.5, 0, (0.1, 0.2, 0.5), 0, 0),
            (-1, 0, -1, 0.5, 2, (1, 1, 1), 0, 1.5), (-1, 0, -1, -0.4, 2, (1, 1, 1), 0, 1.5),
            (1, 0, -1, 0.5, 1, (0.8, 0.6, 0.2), 0.1, 0)]
    for k, (x, y, z, r, kind, alb, fuzz, ior) in enumerate(rows):
        sph[k] = (x, y, z, r)
        mat[k] = (kind, alb, fuzz, ior, (0, 0))
    cam = oracle.make_camera((-2, 2, 1), (0, 0, -1), (0, 1, 0), 40.0, w / h, 0.05, 3.4)
    want, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=12, max_depth=50, seed=1))
    assert np.array_equal(_read_ppm(out), want[::-1, :, :3])


def test_harness_json_metrics(tmp_path):
    """--json 1: one JSON object per frame on stdout (SURVEY section 5's metrics line) and nothing else."""
    import json
    out = str(tmp_path / "m.ppm")
    res = subprocess.run([MAIN, "--scene", "three", "--width", "64", "--height", "36", "--spp", "2", "--frames", "3", "--json", "1",
                          "--out", out], check=True, capture_output=True, text=True)
    lines = [json.loads(x) for x in res.stdout.strip().splitlines()]
    summary = lines.pop()  # bench.py's fields for this path, after the last frame
    assert summary["n_gpus"] == 1 and summary["steps"] == 3 and summary["warmup"] == 0 and summary["ms_per_step"] > 0
    assert summary["config"]["transport"] == "single" and len(summary["config"]["device_kernel_ms"]) == 1
    assert [x["frame"] for x in lines] == [0, 1, 2]
    for x in lines:
        assert x["width"] == 64 and x["height"] == 36 and x["spp"] == 2 and x["spheres"] == 5 and x["gpus"] == 1
        assert x["kernel_ms"] > 0 and x["segments"] >= 64 * 36 * 2 and x["mray_per_s_nominal"] > 0


def test_library_before_torch_in_one_process():
    """Loading librtiow_hip.so (and creating a context) before torch first touches the GPU must leave one
    HIP runtime in the process: the binding preloads the copy of libamdhip64.so that torch ships."""
    code = ("import vulkan_rtiow_amd as V\n"
            "c = V.Context(0)\n"
            "img = c.render_ubo(V.ubo_from_image(64, 48), V.RT_MODE_CH06)\n"
            "import torch\n"
            "x = torch.ones(8, device='cuda:0')\n"
            "print('ok', int(x.sum().item()), img.shape)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok 8 (48, 64, 4)" in res.stdout, res.stderr[-2000:]
