"""CPU: the product's host side (scene builders, camera, tiling, writer, ABI surface)
against the oracle's independent restatement.  No compute entry point is called."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import vulkan_rtiow_amd as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rtiow.h")).read()
    declared = set(re.findall(r"\b(rt[A-Z]\w+)\s*\(", header))
    assert declared == set(V.api.SIGNATURES), declared ^ set(V.api.SIGNATURES)
    lib = V.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rtAbiVersion() == 4
    # ... and NOTHING else (crc32_update, put_be32, write_chunk and the rtiow:: / std:: internals used to be visible to, and
    # interposable by, the application the library is linked into): csrc/rtiow.map
    import subprocess
    for so in ("librtiow_hip.so", "librtiow_hip_knobs.so"):
        path = os.path.join(ROOT, "vulkan-rtiow_amd", so)
        if not os.path.exists(path):
            continue
        out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
        exported = {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}
        assert exported == declared, (so, sorted(exported ^ declared))


def test_chunk_order_fits_its_allocation():
    """the chunk order is stored queue by queue (ceil(n / 8) places per queue), so its last word lies at
    8 * ceil(n / 8) - 2 or - 1 whatever n % 8 is; rtRender must allocate that, not n words.  The functions checked are
    the ones rtRender sizes the buffer with and the kernels index it with."""
    for n in list(range(1, 4100)) + [3750, 30000, 259200, 1 << 20]:
        words, hi = V.chunk_order_selftest_host(n)
        assert hi < words, (n, words, hi)
        assert words == 8 * ((n + 7) // 8)


@pytest.mark.parametrize("kind", ["cover", "cover4096", "layer_x", "cube", "tiny"])
def test_cluster_build_boxes_contain_their_spheres_and_the_flat_interval_contains_the_boxes(kind):
    """rtSetScene's two-level list on the CPU (rtClusterBuildHost runs build_clusters itself): every sphere sits in exactly one
    slot; a small sphere lies inside its cluster's box and the cluster's box inside its super-cluster's; and when the scene
    is flat along an axis -- the cover scenes are, along y -- the common interval contains every box's own interval and the
    flat boxes are the whole boxes with that axis left out.  (Boxes only cull: containment is all correctness needs.)"""
    rng = np.random.default_rng(5)
    if kind in ("cover", "cover4096"):
        sph, _ = V.make_cover_scene(1, 11 if kind == "cover" else 32)
    else:
        n = {"layer_x": 700, "cube": 900, "tiny": 5}[kind]
        sph = np.zeros(n, V.SPHERE_DTYPE)
        pos = rng.uniform(-3, 3, (n, 3))
        if kind == "layer_x":
            pos[:, 0] = rng.uniform(-0.01, 0.01, n)
        sph["cx"], sph["cy"], sph["cz"] = pos[:, 0], pos[:, 1], pos[:, 2]
        sph["radius"] = rng.uniform(0.05, 0.06, n) * rng.choice([1.0, -1.0], n)
    b = V.cluster_build_host(sph)
    idx = b["slot_index"]
    real = idx[idx != 0xFFFFFFFF]
    assert sorted(real.tolist()) == list(range(len(sph)))              # every sphere once
    assert len(idx) == b["n_large_slots"] + 16 * b["n_clusters"]
    c = np.stack([sph["cx"], sph["cy"], sph["cz"]], 1).astype(np.float64)
    r = np.abs(sph["radius"]).astype(np.float64)
    boxes = b["boxes"].astype(np.float64)
    lo, hi = boxes[:, :3] - boxes[:, 3:], boxes[:, :3] + boxes[:, 3:]
    for k in range(b["n_clusters"]):
        members = idx[b["n_large_slots"] + 16 * k: b["n_large_slots"] + 16 * (k + 1)]
        members = members[members != 0xFFFFFFFF]
        for m in members:
            assert (c[m] - r[m] >= lo[k]).all() and (c[m] + r[m] <= hi[k]).all(), (kind, k, m)
        if b["n_super"] and len(members):
            s_ = b["n_clusters"] + k // 8
            assert (lo[k] >= lo[s_]).all() and (hi[k] <= hi[s_]).all(), (kind, k)
    expect_axis = {"cover": 1, "cover4096": 1, "layer_x": 0, "cube": 3, "tiny": None}[kind]
    if expect_axis is not None:
        assert b["flat_axis"] == expect_axis, (kind, b["flat_axis"])
    if b["flat_axis"] < 3:
        fa = b["flat_axis"]
        mid, half = b["flat_interval"]
        # (the CLUSTER boxes: a super-cluster's own interval is the union of its clusters' rounded outwards once more, and what the
        # kernel tests in its place is the common interval with the super's two other axes, which contain its clusters')
        occupied = [k for k in range(b["n_clusters"]) if boxes[k, 3:].max() > 0]   # (padding boxes are points far away)
        assert all(lo[k, fa] >= mid - half and hi[k, fa] <= mid + half for k in occupied), kind
        others = [ax for ax in range(3) if ax != fa]
        want = np.concatenate([b["boxes"][:, others], b["boxes"][:, [3 + others[0], 3 + others[1]]]], 1)
        assert np.array_equal(b["flat_boxes"], want)
        own = (hi[occupied, fa] - lo[occupied, fa]).min()
        assert 2 * half <= 1.5 * own * (1 + 1e-6)                     # never taken when a box is much narrower than the union


@pytest.mark.parametrize("grid_half", [14, 20, 28, 36])
def test_super_clusters_are_subtrees_of_the_build(grid_half):
    """The eight clusters under a super-cluster's box are one subtree of the build's median splits -- the splits of a scene that
    gets super-clusters are rounded to whole super-clusters -- so the super boxes tile the scene: on cover scenes of 785 .. 5185
    spheres (56 .. 328 clusters, none a power of two) their footprints add up to little more than the scene's own.  (Rounded to
    whole clusters only, a super-cluster straddled two subtrees wherever the count was not a power of two: the 3138-sphere
    scene's footprints added up to several times the scene and a ray made 103 tests per segment instead of 59.)"""
    sph, _ = V.make_cover_scene(1, grid_half)
    b = V.cluster_build_host(sph)
    assert b["n_super"] == b["n_clusters"] // 8 and b["n_super"] > 0 and b["flat_axis"] == 1
    sup = b["boxes"][b["n_clusters"]:].astype(np.float64)
    sup = sup[sup[:, 3:].max(axis=1) > 0]
    footprint = (2 * sup[:, 3]) * (2 * sup[:, 5])
    lo = (sup[:, [0, 2]] - sup[:, [3, 5]]).min(axis=0)
    hi = (sup[:, [0, 2]] + sup[:, [3, 5]]).max(axis=0)
    scene = (hi - lo).prod()
    assert footprint.sum() <= 1.35 * scene, (footprint.sum(), scene)
    assert footprint.max() <= 4.0 * scene / len(sup), (footprint.max(), scene / len(sup))


def test_header_is_plain_c_and_links(tmp_path):
    """include/rtiow.h compiles as C99 (plain pointers and sizes, no C++), and a C program that references
    every declared entry point links against librtiow_hip.so — what a cgo / JNI / FFI binding relies on."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    header = open(os.path.join(ROOT, "include", "rtiow.h")).read()
    names = sorted(set(re.findall(r"\b(rt[A-Z]\w+)\s*\(", header)))
    src = tmp_path / "abi.c"
    src.write_text('#include "rtiow.h"\n#include <stddef.h>\nint main(void) {\n    void* fns[] = {'
                   + ", ".join(f"(void*){n}" for n in names) +
                   '};\n    RtParams p; RtStats s; (void)p; (void)s;\n'
                   '    return (sizeof(fns) / sizeof(fns[0]) == %d && sizeof(RtUbo5) == 20 && sizeof(RtSphere) == 16 &&\n'
                   '            sizeof(RtMaterial) == 32 && sizeof(RtCamera) == 88 && sizeof(RtParams) == 56 &&\n'
                   '            rtAbiVersion() == 4 && rtTileRowCount(10, 4, 0, 2) == 6) ? 0 : 1;\n}\n' % len(names))
    libdir = os.path.join(ROOT, "vulkan-rtiow_amd")
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-Wno-pedantic",
                    "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lrtiow_hip",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0      # host-only entry points: no GPU needed


def test_struct_layouts_match_header():
    assert C.sizeof(V.RtUbo5) == 20          # raytrace06.comp:4-10 / main.cpp:109-115
    assert C.sizeof(V.RtSphere) == 16
    assert C.sizeof(V.RtMaterial) == 32
    assert C.sizeof(V.RtCamera) == 88
    assert C.sizeof(V.RtParams) == 56


def test_ubo_formula_matches_reference_main_cpp(oracle):
    for w, h in ((800, 608), (400, 225), (1200, 800), (1, 1), (3840, 2160)):
        a, b = V.ubo_from_image(w, h), oracle.ubo_from_image(w, h)
        assert bytes(a) == bytes(b)
    u = V.ubo_from_image(800, 608)
    assert (u.imageWidth, u.imageHeight, u.viewportWidth, u.focalLength) == (800.0, 608.0, 2.0, 1.0)
    assert u.viewportHeight == np.float32(2.0) / (np.float32(800) / np.float32(608))


def test_cameras_match_oracle(oracle):
    ubo = V.ubo_from_image(400, 225)
    assert bytes(V.camera_from_ubo(ubo)) == bytes(oracle.camera_from_ubo(ubo))
    args = ((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    assert bytes(V.make_camera(*args)) == bytes(oracle.make_camera(*args))
    args = ((-2, 2, 1), (0, 0, -1), (0, 1, 0), 90.0, 16 / 9, 0.0, 1.0)
    assert bytes(V.make_camera(*args)) == bytes(oracle.make_camera(*args))


@pytest.mark.parametrize("seed,half", [(1, 11), (2, 11), (1, 3), (9, 32), (1, 0)])
def test_cover_scene_matches_oracle(oracle, seed, half):
    s1, m1 = V.make_cover_scene(seed, half)
    s2, m2 = oracle.make_cover_scene(seed, half)
    assert s1.tobytes() == s2.tobytes() and m1.tobytes() == m2.tobytes()
    assert len(s1) <= (2 * half) ** 2 + 4


def test_cover_scene_shape():
    sph, mat = V.make_cover_scene(1, 11)
    assert 470 <= len(sph) <= 488            # "~485 spheres"
    assert sph[0]["radius"] == 1000.0 and mat[0]["kind"] == V.RT_MAT_LAMBERTIAN
    kinds = np.bincount(mat["kind"][1:-3], minlength=3) / (len(sph) - 4)
    assert 0.7 < kinds[0] < 0.9 and 0.08 < kinds[1] < 0.22 and 0.01 < kinds[2] < 0.1
    big, _ = V.make_cover_scene(1, 32)
    assert 4000 <= len(big) <= 4100          # BASELINE config 5


def test_three_sphere_scene_matches_oracle(oracle):
    for bubble in (False, True):
        s1, m1 = V.make_three_sphere_scene(bubble)
        s2, m2 = oracle.make_three_sphere_scene(bubble)
        assert s1.tobytes() == s2.tobytes() and m1.tobytes() == m2.tobytes()
        assert len(s1) == 4 + int(bubble)


def test_tile_rows_partition_the_image(oracle):
    for h, block, count in ((800, 16, 8), (803, 16, 8), (225, 4, 2), (7, 3, 4), (2160, 8, 8), (5, 1, 1), (9, 0, 3)):
        seen = []
        for rank in range(count):
            n = V.tile_row_count(h, block, rank, count)
            assert n == oracle.lib.oracle_tile_row_count(h, block, rank, count)
            rows = [V.tile_global_row(lr, block, rank, count) for lr in range(n)]
            assert rows == [oracle.lib.oracle_tile_global_row(lr, block, rank, count) for lr in range(n)]
            assert rows == sorted(rows)
            seen += rows
        assert sorted(seen) == list(range(h))


def test_ppm_writer_matches_oracle(oracle, tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (13, 7, 4), dtype=np.uint8)
    V.write_ppm(str(tmp_path / "a.ppm"), img)
    oracle.write_ppm(str(tmp_path / "b.ppm"), img)
    a, b = (tmp_path / "a.ppm").read_bytes(), (tmp_path / "b.ppm").read_bytes()
    assert a == b and a.startswith(b"P6\n7 13\n255\n")


def _decode_png_rgb(data):
    """Minimal PNG reader for the files rtWritePNG produces (8-bit RGB, filter 0): checks signature and
    every chunk CRC, inflates IDAT with zlib."""
    import struct, zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(typ + body) & 0xFFFFFFFF
        chunks.append((typ, body))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, flt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, flt, lace) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("w,h", [(7, 13), (1, 1), (300, 100)])   # 300x100: 90 100 raw bytes, two stored blocks
def test_png_writer_round_trip(tmp_path, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    V.write_png(str(tmp_path / "a.png"), img)
    got = _decode_png_rgb((tmp_path / "a.png").read_bytes())
    assert np.array_equal(got, img[::-1, :, :3])       # top line of the picture = buffer row H-1 (rt.frag:8)
    lib = V.load_library()
    assert lib.rtWritePNG(b"/nonexistent-dir/x.png", img.ctypes.data, w, h, w * 4) == V.RT_ERR_IO
    assert lib.rtWritePNG(None, img.ctypes.data, w, h, w * 4) == V.RT_ERR_INVALID


def test_error_codes_without_gpu_or_context():
    lib = V.load_library()
    assert lib.rtRender(None, None, None, None, 0, 0, None) == V.RT_ERR_INVALID
    assert lib.rtSetScene(None, None, None, 0) == V.RT_ERR_INVALID
    assert lib.rtUboFromImage(0, 5, C.byref(V.RtUbo5())) == V.RT_ERR_INVALID
    assert lib.rtWritePPM(b"/nonexistent-dir/x.ppm", np.zeros(16, np.uint8).ctypes.data, 2, 2, 8) == V.RT_ERR_IO
    assert lib.rtDestroy(None) == V.RT_OK


@pytest.mark.parametrize("h,w", [(800, 1200), (83, 120), (7, 5), (1, 3), (2160, 64)])
def test_multi_gpu_partition_gather_deinterleave_on_the_host(h, w):
    """the multi-GPU logic of rtMultiRender without a GPU.  rtMultiSelfTestHost cuts a frame
    into the N block-cyclic tiles the devices would render (padding rows poisoned), moves them through memcpy
    standing in for the communicator into the [N][rows_max][width] gather buffer, and puts the rows back with
    the de-interleave the device kernel mirrors.  The frame must come back unchanged for every N and block,
    including tiles that own no row and N above the number of row blocks."""
    rng = np.random.default_rng(h * 131 + w)
    full = rng.integers(0, 2**32, (h, w), dtype=np.uint64).astype(np.uint32)
    for n in (1, 2, 3, 4, 8, 13, 64):
        for block in (0, 1, 4, 16, 1000):
            out = V.multi_selftest_host(full, block, n)
            assert np.array_equal(out, full), (n, block)
            # and the slot size is what dist.py pads its gather to
            if block:
                D = __import__("importlib").import_module("vulkan-rtiow_amd.dist")
                assert D.max_tile_rows(h, block, n) == max(V.tile_row_count(h, block, r, n) for r in range(n))


def test_multi_gpu_entry_points_fail_cleanly_without_a_gpu():
    """No CPU fallback: without a usable device rtCreateMulti reports RT_ERR_NO_DEVICE (on a GPU box the
    -m gpu tests cover the real thing)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by tests/test_gpu_multi.py")
    with pytest.raises(V.RtError) as e:
        V.MultiContext([0, 1])
    assert e.value.code == V.RT_ERR_NO_DEVICE
    with pytest.raises(V.RtError):
        V.MultiContext([])


def test_host_side_under_sanitizers():
    """the host side of the product (scene builders, camera, tiling, PPM/PNG writers, the
    two-level list's build, the scene-file parser fed malformed files) and the oracle, compiled from the product's
    own sources with -fsanitize=address,undefined and run on the CPU (tests/host_asan/).  GPU AddressSanitizer
    is not available on this pool."""
    import shutil
    import subprocess
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no g++ / make")
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "host_asan"), "run"], capture_output=True, text=True,
                         timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "all checks held" in res.stdout
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr


def test_clustered_kernels_keep_their_waves_per_simd():
    """The register budgets the launch configuration relies on (MI355X_MICROARCH: 512 registers per SIMD lane, granule 8),
    from the compiler's own report: the small-scene clustered variants -- the default kernel of the cover scene -- run FOUR
    waves per SIMD (two groups of 512 threads per CU) at <= 128 VGPRs without scratch (round 5; round 4: 4 registers / 20 bytes per
    lane for the variant with whole boxes; round 3: 28 / 60 and 41 / 72, before a path's pixel, entry, line buffer and depth were packed
    into one register and the per-wave LDS areas got one base); the large-scene variants run three at <= 168, or -- COMPACT, round 5 --
    four at <= 128 with a handful of spilled registers; the flat-list kernels five at <= 96 with no scratch.  A change that pushes one
    over would silently cost a quarter or a third of the occupancy."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "vulkan-rtiow_amd", "csrc"), "asm"], capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    text = res.stdout + res.stderr
    vgpr, scratch = {}, {}
    name = None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"\bVGPRs: (\d+)", line)
        if m and name:
            vgpr[name] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch[name] = int(m.group(1))
    # <shading records in LDS, clustered list, flat-axis box test, compact per-wave area>; `make asm` runs all three compilations of
    # rtiow_kernels.hip
    kinds = {k: re.search(r"path_persistent_kernelILb([01])ELb([01])ELb([01])ELb([01])EEEv", k) for k in vgpr}
    kinds = {k: tuple(int(x) for x in m.groups()) for k, m in kinds.items() if m}
    assert sorted(kinds.values()) == [(0, 0, 0, 0), (0, 1, 0, 0), (0, 1, 0, 1), (0, 1, 1, 0), (0, 1, 1, 1), (1, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0)], kinds
    for k, (small, clustered, flat, compact) in kinds.items():
        if small and clustered:
            # (a jump beyond these bounds would say the allocation has tipped over, as it did at 84 spilled registers in round 2
            # and at 28 -- 48.5 MB of scratch write-back per cover frame -- in round 3.  Round 5: no scratch at all, either variant.)
            assert vgpr[k] <= 128 and scratch[k] == 0, (k, vgpr[k], scratch[k])
        elif clustered and compact:
            # (round 5: the large-scene variants at FOUR waves per SIMD -- one 1024-thread group -- possible since the cold kernel
            # arguments left the scalar registers: 4-5 registers / 20 bytes of scratch under the default scheduler, where round 4 spilled 115)
            assert vgpr[k] <= 128 and scratch[k] <= 32, (k, vgpr[k], scratch[k])
        elif clustered:
            # (round 4: a compilation pass of their own under iterative-ilp, which uses all 168 registers three waves allow and may
            # put a few wave-uniform values into scratch)
            assert vgpr[k] <= 168 and scratch[k] <= 32, (k, vgpr[k], scratch[k])
        else:
            assert vgpr[k] <= 96 and scratch[k] == 0, (k, vgpr[k], scratch[k])


def _camera_rays(V, cam, width, height, pix_lo, pix_hi, n, rng, edge=False):
    """n camera rays of the span pix_lo..pix_hi as camera_path samples them (float64; rtiow_kernels.hip); edge: only the
    outermost ones -- corners of the span's first and last pixel, rim of the lens."""
    c = {k: np.array(getattr(cam, k), np.float64) for k in ("origin", "lower_left", "horizontal", "vertical", "u", "v")}
    pix = rng.choice([pix_lo, pix_hi], n) if edge else rng.integers(pix_lo, pix_hi + 1, n)
    i, j = pix % width, pix // width
    # the corners and edges of the pixel as well as its inside, the rim of the lens as well as its disk
    p_edge = 1.0 if edge else 0.3
    xi = np.where(rng.random(n) < p_edge, rng.integers(0, 2, n).astype(float), rng.random(n))
    eta = np.where(rng.random(n) < p_edge, rng.integers(0, 2, n).astype(float), rng.random(n))
    u = (i + xi) / (width - 1)
    v = (j + eta) / (height - 1)
    r = np.where(rng.random(n) < p_edge, 1.0, np.sqrt(rng.random(n))) * float(cam.lens_radius)
    phi = rng.random(n) * 2 * np.pi
    off = (r * np.cos(phi))[:, None] * c["u"] + (r * np.sin(phi))[:, None] * c["v"]
    o = c["origin"] + off
    d = c["lower_left"] + u[:, None] * c["horizontal"] + v[:, None] * c["vertical"] - c["origin"] - off
    return o, d / np.linalg.norm(d, axis=1, keepdims=True)


@pytest.mark.parametrize("case", range(24))
def test_primary_pass_cone_cull_is_conservative(case):
    """The cull of the primary pass (rtiow_kernels.hip, "The primary pass") run on the host through rtConeSelfTestHost --
    the kernels' own cone_of_span / cone_reaches / cone_reaches_sphere -- against brute force: thousands of camera rays
    of a span, sampled in float64 as camera_path samples them (pixel corners and the rim of the lens included), must not
    hit a sphere or enter a box the cull has marked unreachable.  The cull must also cull: most of the scene is out."""
    import vulkan_rtiow_amd as V
    rng = np.random.default_rng(4200 + case)
    width, height = int(rng.integers(2, 400)), int(rng.integers(2, 300))
    scale = float(rng.choice([0.05, 1.0, 1.0, 40.0]))
    frm = rng.normal(size=3) * rng.choice([0.5, 3.0, 12.0]) * scale
    at = rng.normal(size=3) * 0.5 * scale
    aperture = float(rng.choice([0.0, 0.0, 0.1, 0.6])) * scale
    focus = float(np.linalg.norm(frm - at)) * float(rng.uniform(0.3, 1.5)) + 1e-3
    cam = V.make_camera(tuple(frm), tuple(at), (0, 1, 0), float(rng.uniform(5, 150)), width / height, aperture, focus)
    n = 300
    sph = np.zeros(n, V.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"] = (rng.uniform(-8, 8, (3, n)) * scale)
    sph["radius"] = rng.uniform(0.05, 0.6, n) * scale * rng.choice([1.0, 1.0, -1.0, 4.0], n)
    mid = rng.uniform(-8, 8, (200, 3)) * scale
    half = rng.uniform(0.05, 1.5, (200, 3)) * scale
    boxes = np.concatenate([mid, half], axis=1).astype(np.float32)
    row = int(rng.integers(0, height))
    lo = row * width + int(rng.integers(0, width))
    hi = min(lo + int(rng.choice([0, 0, 1, 7, 40])), (row + 1) * width - 1)
    centre = np.zeros(3, np.float32)
    rmax = 2.0 * float(np.linalg.norm([16, 16, 16])) * scale + 2.0 * float(np.linalg.norm(frm))   # the camera is in range
    # ... and tiny spheres ON the outermost rays of the span (pixel corners, rim of the lens), at all distances: the
    # sharpest probes of the footprint and lens bounds (a footprint bound 30 % short fails here)
    ob, db = _camera_rays(V, cam, width, height, lo, hi, 96, rng, edge=True)
    tb = rng.uniform(0.05, 3.0, 96) * focus
    edge = np.zeros(96, V.SPHERE_DTYPE)
    pb = ob + tb[:, None] * db
    edge["cx"], edge["cy"], edge["cz"] = pb[:, 0], pb[:, 1], pb[:, 2]
    edge["radius"] = 1e-4 * tb
    sph = np.concatenate([sph, edge])
    cull, sreach, breach = V.cone_selftest_host(cam, width, height, lo, hi, centre, rmax, sph, boxes)
    assert sreach[n:].all(), (case, np.nonzero(~sreach[n:])[0][:5])
    o, d = _camera_rays(V, cam, width, height, lo, hi, 6000, rng)
    sph = sph[:n]
    sreach = sreach[:n]
    c = np.stack([sph["cx"], sph["cy"], sph["cz"]], axis=1).astype(np.float64)
    r2 = sph["radius"].astype(np.float64) ** 2
    oc = o[:, None, :] - c[None, :, :]
    hb = np.einsum("nsk,nk->ns", oc, d)
    disc = hb * hb - (np.einsum("nsk,nsk->ns", oc, oc) - r2[None, :])
    far_root = -hb + np.sqrt(np.maximum(disc, 0.0))
    hit = ((disc >= 0.0) & (far_root > 0.0)).any(axis=0)          # some ray of the span meets the sphere ahead of its origin
    assert not (hit & ~sreach).any(), (case, np.nonzero(hit & ~sreach)[0][:5])
    b32 = boxes.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t1 = (b32[None, :, :3] - b32[None, :, 3:] - o[:, None, :]) * inv[:, None, :]
        t2 = (b32[None, :, :3] + b32[None, :, 3:] - o[:, None, :]) * inv[:, None, :]
    tn = np.nanmax(np.minimum(t1, t2), axis=2)
    tf = np.nanmin(np.maximum(t1, t2), axis=2)
    enters = ((tf >= np.maximum(tn, 0.0))).any(axis=0)
    assert not (enters & ~breach).any(), (case, np.nonzero(enters & ~breach)[0][:5])
    if cull and hi - lo <= 7 and aperture < 0.5 * scale:
        assert sreach.mean() < 0.6 and breach.mean() < 0.7, (case, sreach.mean(), breach.mean())  # it does cull


def test_a_scene_is_boxed_once_by_rtsetscene():
    """rtSetScene decides a scene's range -- 0.6 scene diagonals with super-clusters, 2 without -- BEFORE it makes a box and
    builds the lists once (round 3 built every scene with super-clusters twice: the one-level range first).  The CPU run of
    its own choice (rtSceneClusterSelfTestHost -> rtiow::build_scene_clusters, the function rtSetScene calls) counts the calls
    of build_clusters."""
    import vulkan_rtiow_amd as V
    for grid, want_range, supers in ((11, 2.0, False), (16, 0.6, True), (32, 0.6, True)):
        sph, _ = V.make_cover_scene(1, grid)
        got = V.scene_cluster_selftest_host(sph)
        assert got["builds"] == 1, (grid, got)
        assert abs(got["range_diags"] - want_range) < 1e-12 and (got["n_super"] != 0) == supers, (grid, len(sph), got)


def test_the_shipped_library_reads_no_debug_environment():
    """The RTIOW_DEBUG_* tuning / test knobs are compiled into the knobs and diagnostic builds only (rtiow_device.h:
    debug_knob): the shipped library holds none of their names -- no environment variable can change its kernel choice, and
    its render path makes no getenv calls -- while the knobs build, which two GPU parity tests and the A/B tools load, holds them."""
    lib = os.path.join(ROOT, "vulkan-rtiow_amd", "librtiow_hip.so")
    knobs = os.path.join(ROOT, "vulkan-rtiow_amd", "librtiow_hip_knobs.so")
    if not (os.path.exists(lib) and os.path.exists(knobs)):
        pytest.skip("libraries not built")
    assert b"RTIOW_DEBUG_" not in open(lib, "rb").read()
    assert b"RTIOW_DEBUG_FLAT" in open(knobs, "rb").read()


def test_fail_loudly_lets_a_clean_exit_through():
    """dist.fail_loudly turns any failure inside it into an immediate os._exit(13) -- but sys.exit(0) is not a failure."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import vulkan_rtiow_amd as V; from importlib import import_module; "
            "d = import_module('vulkan-rtiow_amd.dist')\n"
            "with d.fail_loudly('x'):\n    sys.exit(int(sys.argv[1]))\n") % ROOT
    for want in (0, 3):
        res = subprocess.run([sys.executable, "-c", code, str(want)], capture_output=True, text=True, timeout=300)
        assert res.returncode == want, (want, res.returncode, res.stderr[-500:])


def test_ch_sky_table_is_what_its_generator_derives():
    """csrc/rtiow_ch_sky_table.h -- the sky colour of raytrace06.comp:45-47 as a step function of normalize(dir).y, on which the first
    phase of ch_kernel_rows' two-phase pixels rests -- is exactly what tools/gen_ch_sky_table.py generates: the script evaluates its
    float32 restatement of the shader's sky arithmetic on all 25 million values unit_y + 1 can take, collects the 280 floats where the
    colour changes, and proves on the host -- interval by interval, for every float the kernel's index can send to each bucket -- that
    every float within the guard band of a step is sent to the exact second phase and that the table's colour holds for every float
    within the guard band of an un-flagged one.  (On the GPU, tests/test_gpu_ch_two_phase.py
    finds the same 280 floats by evaluating the kernel's own code on all 2.1 billion floats of [-1, 1].)"""
    import subprocess
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ch_sky_table.py"), "--check"], capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stdout[-500:] + res.stderr[-1500:]
    assert "280 changes in 179 zones" in res.stdout and "691 un-flagged intervals proven" in res.stdout, res.stdout


def test_ch_sky_steps_follow_the_quantiser():
    """An independent look at the step list: the red byte trunc((1 - 0.5 t) * 255 + 0.5) steps where 127.5 t crosses k + 0.5, the green
    byte where 76.5 t does (t = (unit_y + 1) / 2), the blue byte never -- 127 + 76 steps, the coinciding ones (3 (2k + 1) = 5 (2j + 1))
    in one zone; every zone of the generator lies within 2e-6 of such a crossing and every crossing has its zone."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_ch_sky_table", os.path.join(ROOT, "tools", "gen_ch_sky_table.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    table, zones, steps = gen.build()
    red = [(k + 0.5) / 127.5 for k in range(128) if (k + 0.5) / 127.5 <= 1.0]
    green = [(j + 0.5) / 76.5 for j in range(77) if (j + 0.5) / 76.5 <= 1.0]
    crossings = sorted(set(round(2.0 * t - 1.0, 9) for t in red + green))
    merged = [crossings[0]]
    for y in crossings[1:]:
        if y - merged[-1] > 1e-6:
            merged.append(y)
    centres = [0.5 * (float(z[0]) + float(z[1])) for z in zones]
    assert len(merged) == len(zones) == 179
    assert max(abs(a - b) for a, b in zip(merged, centres)) < 2e-6
    assert all((s[1] ^ s[2]) & 0xFF0000 == 0 for s in steps)  # blue never changes
    assert steps[0][1] == 0xFFFFFF and int(gen.F(np.float32(1.0))) & 0xFF00FF == 0xFF0080  # white at the nadir, (0.5, 0.7, 1) at the zenith


def test_camera_precondition_accepts_every_sane_camera_and_refuses_the_rest():
    """rtRender refuses a camera whose rays -- lens to image plane -- could be shorter than 2^-30 or longer than 2^40 (the kernels' short
    square root and reciprocal are exact inside that range; rtCameraIsRenderable is the check, no GPU needed).  Every camera the tests,
    the bench and the harness use passes -- lenses wider than the focus distance, a 179-degree fish-eye, a camera kilometres out -- and
    scenes of picometres or light-days, a non-finite camera and a degenerate image plane do not."""
    ok = [((13, 2, 3), (0, 0, 0), 20.0, 0.1, 10.0), ((13, 2, 3), (0, 0, 0), 20.0, 6.0, 1.0), ((3, 0.6, 2), (0, 0.5, 0), 179.0, 0.0, 1.0),
          ((40, 6, 9), (0, 0, 0), 8.0, 0.0, 40.0), ((0.3, 0.25, 0.4), (4, 0.2, 3), 90.0, 0.05, 2.0), ((4e4, 2e3, 3e4), (0, 0, 0), 1.0, 0.0, 5e4),
          ((1e-3, 2e-3, 3e-3), (0, 0, 0), 40.0, 1e-5, 3.7e-3), ((13, 2, 3), (0, 0, 0), 0.01, 0.0, 10.0),
          ((1e6, 2.0, 3.0), (1e6 + 4, 0, 0), 30.0, 0.0, 4.0)]  # a million units out, focused four units ahead
    for frm, at, fov, ap, focus in ok:
        for aspect in (0.25, 1.5, 8.0):
            assert V.camera_is_renderable(V.make_camera(frm, at, (0, 1, 0), fov, aspect, ap, focus)), (frm, fov, ap, focus, aspect)
    bad = [V.make_camera((13e-12, 2e-12, 3e-12), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.0, 1e-11),
           V.make_camera((13e13, 2e13, 3e13), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.0, 1e14)]
    flat = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    for k in range(3):
        flat.vertical[k] = flat.horizontal[k]
    nan = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    nan.lower_left[1] = float("nan")
    bad.append(V.make_camera((1e6, 2.0, 3.0), (1e6 + 0.5, 0, 0), (0, 1, 0), 30.0, 1.5, 0.0, 0.5))  # ... half a unit: the float ray is noise
    for cam in bad + [flat, nan]:
        assert not V.camera_is_renderable(cam)


def test_blockprof_instruments_the_default_kernel_and_one_without_spare_registers():
    """tools/blockprof (round 5): one scalar atomic in front of every basic block of a kernel's assembly.  On the CPU: the default kernel's
    assembly is instrumented, assembles and links into librtiow_hip_blk.so (hipcc cross-compiles), every block got its counter; so is the
    COMPACT large-scene variant, which has no register to spare."""
    import json
    import shutil
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    build = os.path.join(ROOT, "tools", "blockprof", "build.sh")
    # the COMPACT large-scene variant -- 128 registers in 1024-thread groups: two more would make the launch fail -- is instrumented without
    # them: the counters take the unused lanes of the kernel's scalar-spill register, and the descriptor keeps its register count
    res = subprocess.run(["bash", build, "compact"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "take their place" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
    mc = json.load(open(os.path.join(ROOT, "tools", "blockprof", "_build", "compact_map.json")))
    assert mc["spare_vgprs"][0] == mc["spare_vgprs"][1]
    asm = open(os.path.join(ROOT, "tools", "blockprof", "_build", "rtiow_kernels_blk.s")).read()
    desc = asm[asm.index(".amdhsa_kernel " + mc["kernel"]):]
    assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1)) <= 128
    v = mc["spare_vgprs"][0]
    body = asm[asm.index(mc["kernel"] + ":"):asm.index(".amdhsa_kernel " + mc["kernel"])]
    lanes_used = {int(x) for x in re.findall(r"v_writelane_b32 v%d, s\d+, (\d+)" % v, body)}
    assert {56, 57, 58, 59, 60} <= lanes_used and not (set(range(max(l for l in lanes_used if l < 56) + 1, 56)) & lanes_used)
    res = subprocess.run(["bash", build, "small"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    m = json.load(open(os.path.join(ROOT, "tools", "blockprof", "_build", "small_map.json")))
    assert "path_persistent_kernelILb1ELb1ELb1E" in m["kernel"] and len(m["blocks"]) > 500
    asm = open(os.path.join(ROOT, "tools", "blockprof", "_build", "rtiow_kernels_small_blk.s")).read()
    body = asm[asm.index(m["kernel"] + ":"):]
    body = body[:body.index(".amdhsa_kernel " + m["kernel"])]
    assert body.count("s_atomic_add s2, s[0:1]") == len(m["blocks"])          # one counter per block ...
    offsets = sorted(int(x, 16) for x in re.findall(r"s_atomic_add s2, s\[0:1\], 0x([0-9a-f]+)", body))
    assert offsets == [128 * b for b in range(len(m["blocks"]))]               # ... each on a 128-byte line of its own
    assert sum(len(b) for b in m["blocks"]) > 8000                             # the static instruction lists report.py multiplies them with
    assert os.path.exists(os.path.join(ROOT, "vulkan-rtiow_amd", "librtiow_hip_blk.so"))
    # `lanes`: the same plus the active lanes of every stretch that runs under one EXEC -- a stretch ends at every write of EXEC too, SCC is
    # kept around the population count, and the branches the tripled code pushes out of reach go over trampolines (relax.py assembles)
    bdir = os.path.join(ROOT, "tools", "blockprof", "_build")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "blockprof", "instrument.py"), os.path.join(bdir, "rtiow_kernels_small.s"),
                          os.path.join(bdir, "lanes_test.s"), os.path.join(bdir, "lanes_test_map.json"), "path_persistent_kernelILb1ELb1ELb1E", "280", "3584",
                          "lanes"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    ml = json.load(open(os.path.join(bdir, "lanes_test_map.json")))
    assert ml["lanes"] and len(ml["blocks"]) > len(m["blocks"])
    for blk in ml["blocks"]:  # EXEC is written by a stretch's last instruction only
        assert not any(rest.split(",")[0].strip() in ("exec", "exec_lo", "exec_hi") or "saveexec" in op or op.startswith("v_cmpx") for op, rest, _ in blk[:-1])
    asm = open(os.path.join(bdir, "lanes_test.s")).read()
    assert asm.count("s_bcnt1_i32_b64 s4, exec") == len(ml["blocks"]) == asm.count("s_cselect_b32 s3, 1, 0") == asm.count("s_cmp_lg_u32 s3, 0\n\tv_readlane_b32 s0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "blockprof", "relax.py"), os.path.join(bdir, "lanes_test.s"), os.path.join(bdir, "lanes_test.o")],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "trampolines" in res.stdout, res.stdout[-500:] + res.stderr[-2000:]


def test_a_rebuild_elsewhere_is_the_shipped_library_byte_for_byte(tmp_path):
    """csrc/Makefile names every object's compilation unit id itself (-cuid: hipcc would derive it from the command line, output path included),
    so the library is the same bytes wherever it is built: what `make OUT=/somewhere/else.so` gives is the file the tests, the bench line and the
    profiles ran on."""
    import subprocess
    lib = os.path.join(ROOT, "vulkan-rtiow_amd", "librtiow_hip.so")
    if not os.path.exists("/opt/rocm/bin/hipcc") or not os.path.exists(lib):
        pytest.skip("no hipcc / no built library")
    out = tmp_path / "elsewhere" / "other_name.so"
    res = subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "vulkan-rtiow_amd", "csrc"), f"OUT={out}"], capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert open(out, "rb").read() == open(lib, "rb").read()
