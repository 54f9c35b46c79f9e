"""CPU: pins the oracle against the reference's own result fixtures and the committed
golden vectors (no GPU, no /root/reference access at run time)."""
import json
import os
import zlib

import numpy as np
import pytest

import vulkan_rtiow_amd as V

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _hits(oracle, w, h):
    img = oracle.render_ubo(oracle.ubo_from_image(w, h), V.RT_MODE_CH05)
    return (img[..., 0] == 255) & (img[..., 1] == 0) & (img[..., 2] == 0)


@pytest.mark.parametrize("row", json.load(open(os.path.join(GOLD, "ch_known_answers.json"))),
                         ids=lambda r: f"{r['mode']}_{r['width']}x{r['height']}")
def test_ch_known_answers(oracle, row):
    """raytrace05.comp / raytrace06.comp at the reference's UBO formula."""
    w, h = row["width"], row["height"]
    mode = V.RT_MODE_CH05 if row["mode"] == "CH05" else V.RT_MODE_CH06
    ubo = oracle.ubo_from_image(w, h)
    img = oracle.render_ubo(ubo, mode)
    assert img.shape == (h, w, 4)
    assert img[..., 3].max() == 0  # imageStore(vec4(color,0.0)): raytrace06.comp:66
    hit = _hits(oracle, w, h)
    ys, xs = np.nonzero(hit)
    assert int(hit.sum()) == row["hit_px"]
    assert [int(xs.min()), int(xs.max())] == row["bbox_x"]
    assert [int(ys.min()), int(ys.max())] == row["bbox_y"]
    # colours: the table is robust to +-1 LSB (FMA contraction is the driver's choice)
    probes = {"px00": (0, 0), "pxWH": (h - 1, w - 1), "centre": (h // 2, w // 2),
              "mid_ymin5": (int(ys.min()) + 5, w // 2), "xmin5_mid": (h // 2, int(xs.min()) + 5)}
    for key, (y, x) in probes.items():
        got = img[y, x, :3].astype(int)
        assert np.abs(got - np.array(row[key])).max() <= 1, (key, got, row[key])


def test_ch05_against_reference_jpeg(oracle):
    """The only result fixture the reference holds: RTCHAP05/RTCHAP05/21986.jpg (800x608,
    display orientation = buffer flipped in y by rt.frag:8)."""
    st = json.load(open(os.path.join(GOLD, "ref_jpeg_stats.json")))["RTCHAP05/RTCHAP05/21986.jpg"]
    w, h = st["size"]
    img = oracle.render_ubo(oracle.ubo_from_image(w, h), V.RT_MODE_CH05)[::-1]  # to display rows
    hit = (img[..., 0] == 255) & (img[..., 1] == 0)
    ys, xs = np.nonzero(hit)
    assert int(hit.sum()) == st["red_count"]
    assert [int(xs.min()), int(xs.max())] == st["red_bbox_x"]
    assert [int(ys.min()), int(ys.max())] == st["red_bbox_y"]
    tol = st["jpeg_tolerance"]
    for key, (y, x) in {"top_left": (0, 0), "bottom_left": (h - 1, 0), "top_right": (0, w - 1),
                        "bottom_right": (h - 1, w - 1)}.items():
        assert np.abs(img[y, x, :3].astype(int) - np.array(st[key])).max() <= tol, key


def test_ch05_against_the_stretched_reference_jpeg(oracle):
    """RTCHAP05/RTCHAP05/1728.jpg, the reference's third result fixture: an older state of the program with a
    1024x1024 compute image and UBO {1024, 1024, 2, 2, 1} shown in the 800x600 window, so the sphere is an
    ellipse.  The oracle's 1024^2 frame, flipped like rt.frag:8 and sampled where the window's pixel centres
    fall on the stretched quad (main.cpp:174-179), has the JPEG's silhouette: bbox within 2 px (exact here),
    area within 0.5 % (edge pixels blend under the linear sampler and the JPEG), corner and centre colours
    within JPEG noise.  Pins the UBO path when viewport and window aspect differ."""
    st = json.load(open(os.path.join(GOLD, "ref_jpeg_stats.json")))["RTCHAP05/RTCHAP05/1728.jpg"]
    ubo = V.RtUbo5(*st["ubo"])
    img = oracle.render_ubo(ubo, V.RT_MODE_CH05)[::-1]          # display rows
    cw, ch = st["compute_image"]
    assert img.shape[:2] == (ch, cw)
    w, h = st["size"]
    yy = ((np.arange(h) + 0.5) / h * ch).astype(int)
    xx = ((np.arange(w) + 0.5) / w * cw).astype(int)
    shown = img[yy][:, xx]
    hit = (shown[..., 0] == 255) & (shown[..., 1] == 0)
    ys, xs = np.nonzero(hit)
    tol = st["bbox_tolerance"]
    for got, want in zip((xs.min(), xs.max(), ys.min(), ys.max()), st["red_bbox_x"] + st["red_bbox_y"]):
        assert abs(int(got) - want) <= tol, (got, want)
    assert abs(int(hit.sum()) - st["red_count"]) <= st["count_tolerance"] * st["red_count"]
    jt = st["jpeg_tolerance"]
    for key, (y, x) in {"top_left": (0, 0), "bottom_left": (h - 1, 0), "top_right": (0, w - 1),
                        "bottom_right": (h - 1, w - 1), "centre": (h // 2, w // 2)}.items():
        assert np.abs(shown[y, x, :3].astype(int) - np.array(st[key])).max() <= jt, key


def test_gradient_against_rt01_jpeg(oracle):
    """RT01/RT01/4068.jpg pins the sky gradient's end colours (800x600 window over a 1024^2
    image with viewport 2x2: top/bottom middle of the picture see unit.y = +-1/sqrt(2))."""
    st = json.load(open(os.path.join(GOLD, "ref_jpeg_stats.json")))["RT01/RT01/4068.jpg"]
    ubo = V.RtUbo5(1024.0, 1024.0, 2.0, 2.0, 1.0)
    # RT01's sphere-free kernel == the miss branch; CH06 differs only inside the sphere
    img = oracle.render_ubo(ubo, V.RT_MODE_CH06)[::-1]
    tol = st["jpeg_tolerance"]
    assert np.abs(img[0, 512, :3].astype(int) - np.array(st["top_mid"])).max() <= tol
    assert np.abs(img[1023, 512, :3].astype(int) - np.array(st["bottom_mid"])).max() <= tol


def test_path_regression_crcs(oracle):
    """PATH frames of the oracle itself (parity unpinned by the reference): guards the spec."""
    from golden.make_golden import build_case
    gold = json.load(open(os.path.join(GOLD, "path_oracle_crc.json")))
    for name, g in gold.items():
        sph, mat, cam = build_case(V, oracle, g["scene"], g["width"], g["height"])
        assert len(sph) == g["n_spheres"]
        assert zlib.crc32(sph.tobytes() + mat.tobytes()) == g["scene_crc32"], name
        prm = V.make_params(g["width"], g["height"], spp=g["spp"], max_depth=g["max_depth"],
                            seed=g["seed"], chunk_spp=g["chunk_spp"], quantiser=g["quantiser"])
        img, segs = oracle.render(sph, mat, cam, prm, nthreads=4)
        assert segs == g["segments"], name
        assert zlib.crc32(img.tobytes()) == g["frame_crc32"], name


def test_path_thread_count_and_tiles_do_not_change_pixels(oracle):
    sph, mat = oracle.make_three_sphere_scene(True)
    w, h = 37, 23
    cam = oracle.camera_from_ubo(oracle.ubo_from_image(w, h))
    prm = V.make_params(w, h, spp=3, max_depth=10, seed=5)
    full, segs = oracle.render(sph, mat, cam, prm, nthreads=1)
    full8, segs8 = oracle.render(sph, mat, cam, prm, nthreads=8)
    assert np.array_equal(full, full8) and segs == segs8
    # block-cyclic row tiles reassemble to the same frame (RNG keyed by global pixel)
    for count, block in ((2, 4), (3, 1), (8, 2)):
        frame = np.zeros_like(full)
        total = 0
        for rank in range(count):
            p = V.make_params(w, h, spp=3, max_depth=10, seed=5, row_block=block, tile_rank=rank,
                              tile_count=count)
            part, s = oracle.render(sph, mat, cam, p)
            total += s
            for lr in range(part.shape[0]):
                frame[oracle.lib.oracle_tile_global_row(lr, block, rank, count)] = part[lr]
        assert np.array_equal(frame, full) and total == segs


def test_path_degenerates_to_reference_camera(oracle):
    """With aperture 0, origin 0 the thin-lens camera is raytrace06.comp:53-61's ray generation."""
    ubo = oracle.ubo_from_image(64, 48)
    cam = oracle.camera_from_ubo(ubo)
    assert cam.lens_radius == 0.0
    assert list(cam.lower_left) == [-1.0, -ubo.viewportHeight / 2, -1.0]
    assert list(cam.horizontal) == [2.0, 0.0, 0.0]
    assert list(cam.vertical) == [0.0, ubo.viewportHeight, 0.0]


def test_ppm_is_flipped_like_rt_frag(oracle, tmp_path):
    img = np.zeros((3, 2, 4), np.uint8)
    img[0, :, 0] = 10   # buffer row 0 = scene bottom
    img[2, :, 0] = 30
    p = tmp_path / "o.ppm"
    oracle.write_ppm(str(p), img)
    data = p.read_bytes()
    assert data.startswith(b"P6\n2 3\n255\n")
    body = np.frombuffer(data[len(b"P6\n2 3\n255\n"):], np.uint8).reshape(3, 2, 3)
    assert body[0, 0, 0] == 30 and body[2, 0, 0] == 10


@pytest.mark.parametrize("case", __import__("furnace").CASES, ids=lambda c: c[0])
def test_path_blue_furnace_known_answer(oracle, case):
    """Analytic pin for PATH mode (tests/furnace.py): the blue channel of a single convex sphere under the
    reference's sky is its blue albedo exactly, for any sampling — checked here on the oracle."""
    import furnace
    name, kind, rho_b, fuzz, ior, expected = case
    sph, mat = furnace.scene(kind, rho_b, fuzz, ior)
    w, h = 96, 64
    cam = oracle.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60.0, w / h, 0.0, 1.0)
    for spp, seed in ((1, 1), (7, 99)):
        img, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=spp, max_depth=50, seed=seed))
        furnace.check(img, w, h, expected)
        # bounce limit 1: the one allowed scatter uses it up, every path that hits returns black
        img, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=spp, max_depth=1, seed=seed))
        furnace.check(img, w, h, 0)


@pytest.mark.parametrize("case", __import__("furnace").HEAD_ON, ids=lambda c: c[0])
def test_path_head_on_known_answer(oracle, case):
    """Second analytic pin (tests/furnace.py): the pixel that sees a mirror or glass sphere head-on ends in
    the horizontal sky whatever happens inside — pins normal, reflect and refract against the sky formula."""
    import furnace
    name, kind, albedo, fuzz, ior = case
    sph, mat = furnace.head_on_scene(kind, albedo, fuzz, ior)
    w, h = 513, 385
    cam = oracle.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60.0, w / h, 0.0, 1.0)
    img, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=4, max_depth=50, seed=5))
    furnace.check_head_on(img, kind, albedo)


def test_path_mirror_hall_known_answer(oracle):
    """Third analytic pin (tests/mirrors.py): two mirror balls facing each other over a mirror floor, checked
    against an independent float64 tracer of the deterministic reflection chains (sky, one bounce, ... five)."""
    import mirrors
    w, h = 360, 240
    sph, mat = mirrors.scene()
    img, _ = oracle.render(sph, mat, mirrors.camera(w, h), V.make_params(w, h, spp=8, max_depth=mirrors.MAX_DEPTH, seed=3))
    mirrors.check(img, w, h)


def test_path_blue_white_hall_known_answer(oracle):
    """Fourth analytic pin (tests/furnace.py): several non-absorbing bodies with blue albedo 1, hollow glass
    (negative radius) among them: blue is 255 in every pixel, whatever the paths do."""
    import furnace
    w, h = 120, 80
    sph, mat = furnace.white_hall_scene()
    cam = oracle.make_camera((0, 0.3, 0.6), (0, 0, -1.6), (0, 1, 0), 55.0, w / h, 0.0, 1.0)
    img, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=16, max_depth=furnace.WHITE_HALL_DEPTH, seed=7))
    furnace.check_white_hall(img)


def test_path_glass_ball_known_answer(oracle):
    """Fifth analytic pin (tests/glass.py): the expected colour of a glass ball under the gradient sky, summed in float64
    over the tree of reflect/refract choices and the pixel footprint -- pins Snell, Schlick, the normal flip and the
    choice probabilities by their effect on DIRECTION, which the head-on and furnace pins cannot see."""
    import glass
    w, h = 96, 64
    sph, mat = glass.scene()
    img, _ = oracle.render(sph, mat, glass.camera(w, h), V.make_params(w, h, spp=2048, max_depth=50, seed=3))
    assert glass.check(img, w, h, tol=3) <= 3


def test_path_lambertian_ball_known_answer(oracle):
    """Sixth analytic pin (tests/lambert.py): a lone diffuse ball under the gradient sky has the closed form
    albedo * sky(2/3 n.y) -- cosine-weighted scattering has mean direction 2/3 n and the sky is linear in y.  Pins the
    distribution of the spec's rejection-free unit-vector sampling."""
    import lambert
    w, h = 96, 64
    sph, mat = lambert.scene()
    img, _ = oracle.render(sph, mat, lambert.camera(w, h), V.make_params(w, h, spp=2048, max_depth=50, seed=5))
    assert lambert.check(img, w, h, tol=3) <= 3


def test_path_fuzzy_metal_ball_known_answer(oracle):
    """Seventh analytic pin (tests/fuzzmetal.py): a lone metal ball with fuzz 0.6 -- the expectation is a 2-D integral over
    the fuzz vector (float64 midpoint rule), absorption of directions that point into the surface included."""
    import fuzzmetal
    w, h = 96, 64
    sph, mat = fuzzmetal.scene()
    img, _ = oracle.render(sph, mat, fuzzmetal.camera(w, h), V.make_params(w, h, spp=2048, max_depth=50, seed=5))
    assert fuzzmetal.check(img, w, h, tol=3) <= 3


def test_path_defocused_mirror_ball_known_answer(oracle):
    """Eighth analytic pin (tests/defocus.py): the thin-lens camera.  A mirror ball well out of focus; the expectation is an
    integral over pixel footprint x lens disk (float64, own restatement of the book's camera).  A lens sampled uniformly
    in radius instead of area would be off by 5 bytes."""
    import defocus
    w, h = 96, 64
    sph, mat = defocus.scene()
    img, _ = oracle.render(sph, mat, defocus.camera(w, h), V.make_params(w, h, spp=2048, max_depth=50, seed=5))
    assert defocus.check(img, w, h, tol=3) <= 3



# ---- the reference's compiled shaders as a second witness (tests/golden/make_spirv_witness.py) ----
_SPV = json.load(open(os.path.join(GOLD, "spirv_witness.json")))


@pytest.mark.parametrize("mode", ["CH05", "CH06"])
@pytest.mark.parametrize("size", ["800x608", "400x225"])
def test_oracle_frames_equal_the_reference_spirv_evaluated_in_binary32(oracle, mode, size):
    """raytrace05.comp.spv / raytrace06.comp.spv -- the binaries RTCHAP05/RTCHAP05/main.cpp:136 and RTCHAP06/main.cpp:154 load --
    were executed symbolically and the value `main` stores evaluated operation by operation in binary32 over whole frames
    (generator: tests/golden/make_spirv_witness.py; conventions for what SPIR-V leaves to the driver in the file).  The oracle, which
    restates the GLSL text, must give the same bytes: the one branch without a screenshot -- CH06's normals -- included."""
    fr = _SPV["shaders"][mode]["frames"][size]
    w, h = (int(x) for x in size.split("x"))
    ubo = oracle.ubo_from_image(w, h)
    assert [ubo.imageWidth, ubo.imageHeight, ubo.viewportWidth, ubo.viewportHeight, ubo.focalLength] == fr["ubo"]
    img = oracle.render_ubo(ubo, V.RT_MODE_CH05 if mode == "CH05" else V.RT_MODE_CH06)
    assert img[..., 3].max() == fr["alpha_max"] == 0
    for x, y, r, g, b in fr["samples"]:
        assert img[y, x, :3].tolist() == [r, g, b], (x, y)
    assert zlib.crc32(img.tobytes()) & 0xFFFFFFFF == fr["crc32_rgba8_row0_bottom"]


def _match(nodes, i, pat, binds):
    """pattern: ("Op", sub ...) | "$name" (binds a node number; the same name must meet the same node) | a float constant"""
    n = nodes[i]
    if isinstance(pat, str):
        if pat in binds:
            return binds[pat] == i
        binds[pat] = i
        return True
    if isinstance(pat, float):
        return n[0] == "const" and n[1] == pat
    if n[0] != pat[0] or len(n) != len(pat):
        return False
    return all((_match(nodes, c, p, binds) if isinstance(c, int) and not isinstance(p, int) else c == p) for c, p in zip(n[1:], pat[1:]))


def test_the_operation_order_the_oracle_restates_is_the_spirv_binarys():
    """What oracle/rtiow_oracle.c:174-214 (and the kernels' ch_pixel) restate, checked against the expression trees of the reference's
    binaries instead of "read side by side by a person": b = 2 * dot(oc, dir); c = dot(oc, oc) - r * r; disc = b * b - (4 * a) * c;
    t = (-b - sqrt(disc)) / (2 * a) behind disc < 0 -> -1; CH05 returns disc > 0; u = float(gid.x) / (imageWidth - 1);
    normalize is the extended instruction (no hand-written division); nothing carries NoContraction; 16 x 16 x 1 workgroups."""
    c6, c5 = _SPV["shaders"]["CH06"], _SPV["shaders"]["CH05"]
    for sh in (c5, c6):
        assert sh["local_size"] == [16, 16, 1]                      # raytrace06.comp:2
        assert sh["no_contraction_decorations"] == 0                # contraction is the driver's choice: parity to +-1 LSB only
        assert sh["float_constants"] == [-1.0, 0.0, 0.5, 0.699999988079071, 1.0, 2.0, 4.0]
    oc = ("vec", ("FSub", "$ox", "$cx"), ("FSub", "$oy", "$cy"), ("FSub", "$oz", "$cz"))
    b = ("FMul", 2.0, ("Dot", "$oc", "$dir"))
    disc = ("FSub", ("FMul", "$b", "$b"), ("FMul", ("FMul", 4.0, ("Dot", "$dir", "$dir")), ("FSub", ("Dot", "$oc", "$oc"), ("FMul", "$r", "$r"))))
    n6 = c6["expression_nodes"]
    binds = {}
    t = ("Select", ("FOrdLessThan", "$disc", 0.0), -1.0, ("FDiv", ("FSub", ("FNegate", "$b"), ("Sqrt", "$disc")), ("FMul", 2.0, ("Dot", "$dir", "$dir"))))
    assert _match(n6, c6["functions"]["hitSphere"]["returns"], t, binds)
    assert _match(n6, binds["$disc"], disc, binds) and _match(n6, binds["$b"], b, binds) and _match(n6, binds["$oc"], oc, binds)
    assert c6["functions"]["hitSphere"]["ext_insts"] == ["Sqrt"] and c6["functions"]["rayColor"]["ext_insts"] == ["Normalize"]
    n5 = c5["expression_nodes"]
    binds = {}
    assert _match(n5, c5["functions"]["hitSphere"]["returns"], ("FOrdGreaterThan", "$disc", 0.0), binds)   # raytrace05.comp:29
    assert _match(n5, binds["$disc"], disc, binds) and _match(n5, binds["$b"], b, binds)
    assert c5["functions"]["hitSphere"]["ext_insts"] == []
    for sh, nodes in ((c5, n5), (c6, n6)):                          # raytrace06.comp:57-58
        uses = [n for n in nodes if n[0] == "FDiv" and nodes[n[1]][0] == "ConvertUToF"]
        assert len(uses) == 2
        for n in uses:
            den, gid = nodes[n[2]], nodes[nodes[n[1]][1]]
            assert den[0] == "FSub" and nodes[den[1]] == ["ubo", gid[1]] and nodes[den[2]] == ["const", 1.0]
        # the stored texel: vec4(colour, 0.0) at (gid.x, gid.y)
        st = sh["functions"]["main"]["stores"]
        assert nodes[nodes[st["colour"]][4]] == ["const", 0.0]
    # CH06's hit colour is 0.5 * (N + 1) with N = normalize(orig + dir * t - (0, 0, -1)): the sky and the normal branch under t > 0
    col = n6[c6["functions"]["rayColor"]["returns"]]
    assert col[0] == "vec" and all(n6[k][0] == "Select" and n6[n6[k][1]][0] == "FOrdGreaterThan" for k in col[1:])
    hit_x = n6[n6[col[1]][2]]
    # (VectorTimesScalar: the vector's component first, then the scalar)
    assert hit_x[0] == "FMul" and n6[hit_x[2]] == ["const", 0.5] and n6[hit_x[1]][0] == "FAdd" and n6[n6[hit_x[1]][1]][0] == "NormalizeComponent"
