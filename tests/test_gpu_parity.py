"""GPU (-m gpu): the HIP path through the C ABI against the CPU oracle, bit for bit."""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

import vulkan_rtiow_amd as V

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KERNELS = [V.KERNEL_PIXEL, V.KERNEL_PERSISTENT, V.KERNEL_CLUSTERED, V.KERNEL_CLUSTERED_PASS]


def _diff(a, b):
    d = np.abs(a.astype(int) - b.astype(int))
    return int(d.max()), int((d > 0).sum())


# ---- arithmetic contract -------------------------------------------------------
@pytest.mark.parametrize("op,name", [(0, "fma"), (1, "div"), (2, "sqrt"), (3, "mul"), (4, "add"), (5, "rng"),
                                     (6, "fixed_accumulate"), (7, "u64_to_float")])
def test_arith_bit_exact(gpu_ctx, oracle, op, name):
    """fma / divide / sqrt / RNG on gfx950 == x86, including denormals, zeros and huge values."""
    rng = np.random.default_rng(op)
    n = 1 << 16
    bits = rng.integers(0, 2**32, (3, n), dtype=np.uint64).astype(np.uint32)
    a, b, c = (x.view(np.float32).copy() for x in bits)
    special = np.array([0.0, -0.0, 1.0, -1.0, 1e-45, -1e-45, 1.1754942e-38, 3.4e38, 1e-20, 0.001,
                        0.999, 255.0, 1e6, 2.0 ** -126, 2.0 ** -149, 4.0, 0.25], np.float32)
    k = len(special)
    a[:k * k] = np.repeat(special, k)
    b[:k * k] = np.tile(special, k)
    c[:k] = special
    if op == 2:
        a = np.abs(a)
    if op == 6:   # radiance-like magnitudes around the clamp points as well as random bit patterns
        a[k * k:k * k + 4096] = rng.random(4096, dtype=np.float32) * np.float32(1.5)
        b[k * k:k * k + 4096] = rng.random(4096, dtype=np.float32) * np.float32(40000.0)
    if op == 7:   # raw 64-bit patterns: keep a and b as they are (NaN bit patterns are just integers here)
        a, b = bits[0].view(np.float32).copy(), bits[1].view(np.float32).copy()
    if op != 7:
        for arr in (a, b, c):  # NaN/inf payloads are outside the contract
            arr[~np.isfinite(arr)] = 1.5
    if op == 6:   # ... and what a broken path may hand the accumulator: NaN of either sign -> 0, +inf -> the clamp, -inf -> 0
        a[6000:6008] = np.array([np.nan, -np.nan, np.inf, -np.inf, np.nan, 3.0e38, -3.0e38, np.inf], np.float32)
        b[6000:6008] = np.array([0.25, np.nan, 0.5, np.inf, np.nan, np.inf, np.nan, -np.inf], np.float32)
    got = gpu_ctx.selftest_arith(op, a, b, c)
    want = oracle.arith(op, a, b, c)
    ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert ok.all(), (name, int((~ok).sum()), a[~ok][:4], b[~ok][:4], got[~ok][:4], want[~ok][:4])


def test_lean_square_root_and_quotient_equal_the_correctly_rounded_ones(gpu_ctx, oracle):
    """The kernels take most of their square roots and the quotient of unit3 by the cores of hipcc's correctly rounded expansions,
    without the range handling around them (rtiow_kernels.hip: lean_sqrt, lean_div).  On operands inside the stated
    preconditions -- x == 0 or x >= 2^-96 (any magnitude above), NaN, +inf, negatives; normal a and b less than 96 binades apart with a
    normal quotient, a == 0 -- they are the oracle's sqrtf and / bit for bit: 2^18 random operands each, plus the edges."""
    rng = np.random.default_rng(404)
    n = 1 << 18
    # square root: exponents from 2^-96 up to the top, every mantissa pattern
    e = rng.integers(127 - 96, 255, n, dtype=np.uint64)
    m = rng.integers(0, 1 << 23, n, dtype=np.uint64)
    x = ((e << 23) | m).astype(np.uint32).view(np.float32).copy()
    edges = np.array([0.0, -0.0, 2.0 ** -96, np.float32(2.0 ** -96) * np.float32(1.0000001), 1.0, 4.0, 2.0, 3.4028235e38, np.inf, np.nan,
                      -1.0, -np.inf, 1e-20, 0.25, 0.99999994, 1.0000001], np.float32)
    x[:len(edges)] = edges
    dummy = np.ones(n, np.float32)
    got = gpu_ctx.selftest_arith(8, x, dummy, dummy)
    want = oracle.arith(2, x, dummy, dummy)
    ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert ok.all(), ("lean_sqrt", int((~ok).sum()), x[~ok][:4], got[~ok][:4], want[~ok][:4])
    # quotient: a in [2^-40, 2^40] with either sign (and 0), b in [2^-40, 2^40]: exponents at most 80 apart, the quotient normal
    def pick(lo, hi):
        ee = rng.integers(127 + lo, 127 + hi, n, dtype=np.uint64)
        mm = rng.integers(0, 1 << 23, n, dtype=np.uint64)
        sg = rng.integers(0, 2, n, dtype=np.uint64)
        return ((sg << 31) | (ee << 23) | mm).astype(np.uint32).view(np.float32).copy()
    a, b = pick(-40, 40), pick(-40, 40)
    a[:8] = np.array([0.0, 1.0, 1.0, 1.0, -1.0, 3.0, 1.0, 2.0 ** 40], np.float32)
    b[:8] = np.array([3.0, 3.0, 1e-8, 4.0, 7.0, 1.0, 0.99999994, 2.0 ** -40], np.float32)
    got = gpu_ctx.selftest_arith(9, a, b, dummy)
    want = oracle.arith(1, a, b, dummy)
    ok = got.view(np.uint32) == want.view(np.uint32)
    ok |= (got == 0.0) & (want == 0.0)  # (a zero dividend: the lean quotient is +0 where IEEE gives -0 for a negative divisor; no kernel site can tell)
    assert ok.all(), ("lean_div", int((~ok).sum()), a[~ok][:4], b[~ok][:4], got[~ok][:4], want[~ok][:4])
    # unit3_scattered's own use: 1 / sqrt(x) for x in [1e-16, 16]
    xs = (rng.random(n, dtype=np.float32) * np.float32(16.0)).astype(np.float32) + np.float32(1e-16)
    r = gpu_ctx.selftest_arith(8, xs, dummy, dummy)
    got = gpu_ctx.selftest_arith(9, dummy, r, dummy)
    want = oracle.arith(1, dummy, oracle.arith(2, xs, dummy, dummy), dummy)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---- the reference's own kernels -------------------------------------------------
@pytest.mark.parametrize("mode", [V.RT_MODE_CH05, V.RT_MODE_CH06])
@pytest.mark.parametrize("w,h", [(800, 608), (400, 225), (1024, 1024), (33, 17), (16, 16), (1, 1), (1200, 800)])
def test_ch_kernels_bit_exact(gpu_ctx, oracle, mode, w, h):
    """raytrace05.comp / raytrace06.comp incl. ragged sizes the reference's floor-div/over-dispatch
    would have mishandled (RTCHAP05 main.cpp:306, RTCHAP06 main.cpp:321)."""
    ubo = oracle.ubo_from_image(w, h)
    want = oracle.render_ubo(ubo, mode)
    got = gpu_ctx.render_ubo(ubo, mode)
    assert got.shape == want.shape
    assert np.array_equal(got, want), _diff(got, want)
    st = gpu_ctx.stats()
    assert st.bytes_written == want.shape[0] * want.shape[1] * 4 and st.kernel_ms > 0


@pytest.mark.parametrize("mode", [V.RT_MODE_CH05, V.RT_MODE_CH06])
def test_ch_row_kernel_equals_the_tiled_form_and_the_oracle(gpu_ctx, oracle, mode):
    """ch_kernel_rows (four pixels of a row per lane, per-column / per-row terms hoisted into LDS, 16-byte stores) against
    ch_kernel_tiles (RtParams.kernel = 1: one lane per pixel in 16x16 tiles, statement for statement raytrace06.comp) and
    against the oracle: widths around the 256-column tile and the 4-pixel quad, heights around the rows of a workgroup,
    one-row and one-column images (u or v is 0/0: the tiled form takes them), and a device destination whose pitch makes
    the rows 4- but not 16-byte aligned (scalar stores instead of the 16-byte one)."""
    import torch
    for w, h in ((255, 3), (256, 4), (257, 5), (258, 2), (2, 2), (3, 2), (5, 1), (1, 5), (1023, 7), (1027, 65), (1200, 800),
                 (4099, 130)):
        rows = gpu_ctx.render(None, V.make_params(w, h, mode=mode))
        tiles = gpu_ctx.render(None, V.make_params(w, h, mode=mode, kernel=V.KERNEL_PIXEL))
        want = oracle.render_ubo(oracle.ubo_from_image(w, h), mode)
        assert np.array_equal(rows, tiles), (w, h, _diff(rows, tiles))
        assert np.array_equal(rows, want), (w, h, _diff(rows, want))
    w, h = 1026, 33
    want = oracle.render_ubo(oracle.ubo_from_image(w, h), mode)
    for pad in (1, 2, 3, 4):  # pitch = (w + pad) words
        buf = torch.full((h, w + pad), 0x55555555, dtype=torch.int32, device="cuda")
        gpu_ctx.render_device(None, V.make_params(w, h, mode=mode), buf.data_ptr(), (w + pad) * 4, 0)
        gpu_ctx.synchronize()
        got = buf.cpu().numpy()
        assert (got[:, w:] == 0x55555555).all(), pad  # nothing written past the row
        assert np.array_equal(got[:, :w].copy().view(np.uint8).reshape(h, w, 4), want), pad


@pytest.mark.parametrize("mode", [V.RT_MODE_CH05, V.RT_MODE_CH06])
def test_ch_lean_roots_and_quotients_equal_the_full_forms_and_the_oracle(gpu_ctx, knobs_ctx, oracle, mode, monkeypatch):
    """ch_kernel_rows takes its square roots and quotients without hipcc's range handling (ch_sqrt, ch_div: the cores of
    the correctly rounded expansions) when the camera's proportions are moderate, and with the full forms otherwise
    (RTIOW_DEBUG_CH_FULL forces those -- in the knobs build of the library: the shipped one reads no environment): random UBOs -- viewports and focal lengths over six orders of magnitude, the
    sphere anywhere from a dot to larger than the frame -- must give the oracle's bytes either way; so must UBOs outside the
    lean range (tiny, huge and negative viewports)."""
    rng = np.random.default_rng(mode)
    ubos = []
    for _ in range(24):
        w, h = int(rng.integers(2, 700)), int(rng.integers(2, 300))
        u = V.ubo_from_image(w, h)
        u.viewportWidth = float(np.float32(10.0 ** rng.uniform(-3, 3)))
        u.viewportHeight = float(np.float32(10.0 ** rng.uniform(-3, 3)))
        u.focalLength = float(np.float32(10.0 ** rng.uniform(-3, 3)))
        ubos.append(u)
    # ... and UBOs at the corners of the lean range, where a unit vector's component IS +-1 and a colour channel 0 or 1 to the last
    # bit (the lean kernels' quantiser has no clamp: ch_unorm8<true>)
    for vw, vh, f in ((2.0 ** -19, 2.0 ** 19, 2.0 ** -19), (2.0 ** 19, 2.0 ** -19, 2.0 ** -19), (2.0 ** -19, 2.0 ** -19, 2.0 ** 19),
                      (2.0 ** 19, 2.0 ** 19, 2.0 ** 19), (1.5, 2.0 ** 19, 0.75), (2.0 ** 19, 1.0, 2.0 ** -19)):
        u = V.ubo_from_image(97, 61)
        u.viewportWidth, u.viewportHeight, u.focalLength = vw, vh, f
        ubos.append(u)
    for vw, vh, f in ((1e-9, 2.0, 1.0), (2.0, 3e8, 1.0), (2.0, 1.0, 1e-12), (-2.0, 1.125, 1.0), (2.0, -1.0, 5e7)):
        u = V.ubo_from_image(96, 64)
        u.viewportWidth, u.viewportHeight, u.focalLength = vw, vh, f
        ubos.append(u)
    for u in ubos:
        want = oracle.render_ubo(u, mode)
        monkeypatch.setenv("RTIOW_DEBUG_CH_FULL", "1")
        lean = gpu_ctx.render_ubo(u, mode)    # (the shipped library: the variable means nothing to it)
        full = knobs_ctx.render_ubo(u, mode)  # (the knobs build: hipcc's full forms)
        monkeypatch.delenv("RTIOW_DEBUG_CH_FULL", raising=False)
        assert np.array_equal(knobs_ctx.render_ubo(u, mode), lean)  # (same kernels without the knob)
        what = (u.imageWidth, u.imageHeight, u.viewportWidth, u.viewportHeight, u.focalLength)
        assert np.array_equal(full, want), (what, _diff(full, want))
        assert np.array_equal(lean, want), (what, _diff(lean, want))


def test_the_short_square_root_and_reciprocal_are_the_compilers_on_every_float(gpu_ctx):
    """The kernels' square root (newton_sqrt: v_rsq_f32 and one Newton step, six instructions instead of the compiler's sixteen) is a
    function of one float, so its claim -- equal to sqrtf for x == 0 and every x in [2^-96, FLT_MAX] -- is checked on EVERY such float:
    1 879 048 192 of them on the GPU (rtSelfTestUnaryScan), zero mismatches -- and likewise the reciprocal (newton_rcp: v_rcp_f32 and one
    Newton step, three instructions instead of twelve) against 1.0f / x on the 1 073 741 824 floats of [2^-64, 2^64); sqrtf itself (op 2) and the form (op 17) against the host's
    correctly rounded root on samples; and the special operands it is used with."""
    bad, first = gpu_ctx.selftest_unary_scan(0, 2.0 ** -96, float(np.finfo(np.float32).max))
    assert bad == 0, (bad, first)
    bad, first = gpu_ctx.selftest_unary_scan(1, 2.0 ** -64, 2.0 ** 64)
    assert bad == 0, (bad, first)
    rng = np.random.default_rng(17)
    x = np.concatenate([(rng.uniform(1.0, 2.0, 1 << 18) * 2.0 ** rng.integers(-96, 127, 1 << 18)).astype(np.float32),
                        np.array([0.0, -0.0, 2.0 ** -96, np.finfo(np.float32).max, 1.0, 0.25, 2.0], np.float32)])
    z = np.zeros_like(x)
    got, full = gpu_ctx.selftest_arith(17, x, z, z), gpu_ctx.selftest_arith(2, x, z, z)
    want = np.sqrt(x.astype(np.float64)).astype(np.float32)  # (double rounding cannot bite: a float's root is never that close to a tie)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.isnan(gpu_ctx.selftest_arith(17, np.array([np.nan], np.float32), z[:1], z[:1])[0])
    s = (rng.uniform(1.0, 2.0, 1 << 18) * 2.0 ** rng.integers(-64, 64, 1 << 18)).astype(np.float32)
    zs = np.zeros_like(s)
    assert np.array_equal(gpu_ctx.selftest_arith(18, s, zs, zs).view(np.uint32), (np.float32(1.0) / s).view(np.uint32))
    # below 2^-96 it may be an ulp off (the residual underflows), like the compiler's core without its rescaling -- but never far off:
    # the ray-sphere tests' argument needs a value within a few ulps of the root there
    tiny = (rng.uniform(1.0, 2.0, 1 << 16) * 2.0 ** rng.integers(-126, -96, 1 << 16)).astype(np.float32)
    t_got = gpu_ctx.selftest_arith(17, tiny, z[: tiny.size], z[: tiny.size]).astype(np.float64)
    assert np.all(np.abs(t_got / np.sqrt(tiny.astype(np.float64)) - 1.0) < 2.0 ** -21)


def test_ch_known_answers_on_gpu(gpu_ctx):
    """SURVEY 8(c) table straight against the HIP kernel (no oracle in the loop)."""
    for row in json.load(open(os.path.join(GOLD, "ch_known_answers.json"))):
        w, h = row["width"], row["height"]
        ubo = V.ubo_from_image(w, h)
        img5 = gpu_ctx.render_ubo(ubo, V.RT_MODE_CH05)
        hit = (img5[..., 0] == 255) & (img5[..., 1] == 0)
        ys, xs = np.nonzero(hit)
        assert int(hit.sum()) == row["hit_px"]
        assert [int(xs.min()), int(xs.max())] == row["bbox_x"] and [int(ys.min()), int(ys.max())] == row["bbox_y"]
        img = img5 if row["mode"] == "CH05" else gpu_ctx.render_ubo(ubo, V.RT_MODE_CH06)
        assert img[..., 3].max() == 0
        for key, (y, x) in {"px00": (0, 0), "pxWH": (h - 1, w - 1), "centre": (h // 2, w // 2)}.items():
            assert np.abs(img[y, x, :3].astype(int) - np.array(row[key])).max() <= 1


def test_ch_via_rtrender_and_custom_ubo(gpu_ctx, oracle):
    prm = V.make_params(400, 225, mode=V.RT_MODE_CH06)
    got = gpu_ctx.render(None, prm)
    assert np.array_equal(got, oracle.render_ubo(oracle.ubo_from_image(400, 225), V.RT_MODE_CH06))
    ubo = V.RtUbo5(321.0, 123.0, 3.5, 2.0, 0.7)  # the book's "viewport height = 2" style values
    assert np.array_equal(gpu_ctx.render_ubo(ubo, V.RT_MODE_CH06), oracle.render_ubo(ubo, V.RT_MODE_CH06))


# ---- PATH mode -------------------------------------------------------------------
def _case(oracle, scene, w, h):
    from golden.make_golden import build_case
    return build_case(V, oracle, scene, w, h)


@pytest.mark.parametrize("kernel", KERNELS)
def test_path_golden_crcs_on_gpu(gpu_ctx, oracle, kernel):
    """Committed oracle CRCs (tests/golden/path_oracle_crc.json) reproduced by the HIP path."""
    gold = json.load(open(os.path.join(GOLD, "path_oracle_crc.json")))
    for name, g in gold.items():
        sph, mat, cam = _case(oracle, g["scene"], g["width"], g["height"])
        gpu_ctx.set_scene(sph, mat)
        prm = V.make_params(g["width"], g["height"], spp=g["spp"], max_depth=g["max_depth"], seed=g["seed"],
                            chunk_spp=g["chunk_spp"], quantiser=g["quantiser"], kernel=kernel)
        img = gpu_ctx.render(cam, prm)
        st = gpu_ctx.stats()
        assert zlib.crc32(img.tobytes()) == g["frame_crc32"], name
        assert st.segments == g["segments"], name
        assert st.paths == g["width"] * g["height"] * g["spp"]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("scene,w,h,spp,depth,chunk,quant", [
    ("three", 400, 225, 8, 50, 0, V.RT_QUANT_BOOK),          # BASELINE config 2 shape, fewer spp
    ("three_bubble", 131, 77, 6, 50, 4, V.RT_QUANT_UNORM8),  # negative radius + ragged chunks
    ("cover11", 150, 100, 4, 50, 0, V.RT_QUANT_BOOK),        # BASELINE config 3 scene
    ("cover11", 96, 64, 9, 5, 2, V.RT_QUANT_BOOK),           # depth exhaustion (black paths)
    ("cover32", 64, 40, 2, 50, 1, V.RT_QUANT_BOOK),          # ~4000 spheres: BASELINE config 5 scene
    ("cover3", 17, 9, 3, 50, 0, V.RT_QUANT_BOOK),            # image smaller than one tile
    ("cover39", 40, 24, 1, 50, 0, V.RT_QUANT_BOOK),          # ~6000 spheres: the largest list LDS holds
])
def test_path_bit_exact_vs_oracle(gpu_ctx, oracle, kernel, scene, w, h, spp, depth, chunk, quant):
    sph, mat, cam = _case(oracle, scene, w, h)
    gpu_ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=11, chunk_spp=chunk, quantiser=quant, kernel=kernel)
    want, segs = oracle.render(sph, mat, cam, prm)
    got = gpu_ctx.render(cam, prm)
    assert np.array_equal(got, want), _diff(got, want)   # stronger than the L_inf <= 1/255 bar
    st = gpu_ctx.stats()
    assert st.segments == segs
    if kernel not in (V.KERNEL_CLUSTERED, V.KERNEL_CLUSTERED_PASS):
        assert st.sphere_tests == segs * len(sph)
    else:   # the two-level list must test fewer spheres than the flat one on a real scene
        assert st.sphere_tests > 0 and (len(sph) < 64 or st.sphere_tests < segs * len(sph))
    assert got[..., 3].max() == 0


@pytest.mark.parametrize("kernel", KERNELS)
def test_path_tiles_reassemble_to_the_single_gpu_frame(gpu_ctx, oracle, kernel):
    """1/2/4/8-GPU frames are byte-identical by construction: RNG keyed by the global pixel."""
    w, h = 120, 83
    sph, mat, cam = _case(oracle, "cover11", w, h)
    gpu_ctx.set_scene(sph, mat)
    base = V.make_params(w, h, spp=3, max_depth=20, seed=2, kernel=kernel)
    full = gpu_ctx.render(cam, base)
    for count, block in ((2, 16), (4, 8), (8, 4), (3, 1)):
        frame = np.zeros_like(full)
        for rank in range(count):
            prm = V.make_params(w, h, spp=3, max_depth=20, seed=2, row_block=block, tile_rank=rank,
                                tile_count=count, kernel=kernel)
            part = gpu_ctx.render(cam, prm)
            assert part.shape[0] == V.tile_row_count(h, block, rank, count)
            for lr in range(part.shape[0]):
                frame[V.tile_global_row(lr, block, rank, count)] = part[lr]
        assert np.array_equal(frame, full), (count, block)


@pytest.mark.parametrize("kernel", KERNELS)
def test_progressive_accumulation_equals_one_big_dispatch(gpu_ctx, oracle, kernel):
    """The frame loop with a running average (RTCHAP06/main.cpp:304-360 re-dispatches every frame):
    k dispatches of spp samples each == one dispatch of k*spp samples, bit for bit, and every
    intermediate average is the oracle's frame at that sample count."""
    w, h = 61, 37
    sph, mat, cam = _case(oracle, "cover11", w, h)
    gpu_ctx.set_scene(sph, mat)
    done = 0
    for spp in (3, 1, 4, 2):
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=4, kernel=kernel, sample_offset=done, accumulate=1)
        got = gpu_ctx.render(cam, prm)
        done += spp
        want, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=done, max_depth=50, seed=4))
        assert np.array_equal(got, want), done
    # a dispatch that does not continue where the accumulators stand is refused
    with pytest.raises(V.RtError) as e:
        gpu_ctx.render(cam, V.make_params(w, h, spp=2, seed=4, kernel=kernel, sample_offset=done + 5, accumulate=1))
    assert e.value.code == V.RT_ERR_STATE
    # restarting at offset 0 resets them
    again = gpu_ctx.render(cam, V.make_params(w, h, spp=3, max_depth=50, seed=4, kernel=kernel, accumulate=1))
    want3, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=3, max_depth=50, seed=4))
    assert np.array_equal(again, want3)


def test_path_seed_and_determinism(gpu_ctx, oracle):
    sph, mat, cam = _case(oracle, "three", 64, 36)
    gpu_ctx.set_scene(sph, mat)
    a = gpu_ctx.render(cam, V.make_params(64, 36, spp=4, seed=1))
    b = gpu_ctx.render(cam, V.make_params(64, 36, spp=4, seed=1))
    c = gpu_ctx.render(cam, V.make_params(64, 36, spp=4, seed=2))
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_device_destination_and_pitch(gpu_ctx, oracle):
    torch = pytest.importorskip("torch")
    w, h = 50, 20
    sph, mat, cam = _case(oracle, "three", w, h)
    gpu_ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=2, seed=3)
    want = gpu_ctx.render(cam, prm)
    pitch_px = 64
    buf = torch.full((h, pitch_px), 0x7F7F7F7F, dtype=torch.int32, device="cuda:0")
    ts = torch.cuda.Stream()
    ts.wait_stream(torch.cuda.current_stream())   # buf's fill
    with torch.cuda.stream(ts):
        gpu_ctx.render_device(cam, prm, buf.data_ptr(), pitch_px * 4, ts.cuda_stream)
    ts.synchronize()
    assert gpu_ctx.stats().kernel_ms > 0
    host = buf.cpu().numpy().view(np.uint8).reshape(h, pitch_px, 4)
    assert np.array_equal(host[:, :w], want)
    assert (host[:, w:] == 0x7F).all()   # padding untouched


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("case", __import__("furnace").CASES, ids=lambda c: c[0])
def test_path_blue_furnace_known_answer_on_gpu(gpu_ctx, case, kernel):
    """The oracle-independent analytic pin of tests/furnace.py, on every PATH kernel."""
    import furnace
    name, kind, rho_b, fuzz, ior, expected = case
    sph, mat = furnace.scene(kind, rho_b, fuzz, ior)
    w, h = 96, 64
    cam = V.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60.0, w / h, 0.0, 1.0)
    gpu_ctx.set_scene(sph, mat)
    for spp, seed in ((1, 1), (7, 99)):
        img = gpu_ctx.render(cam, V.make_params(w, h, spp=spp, max_depth=50, seed=seed, kernel=kernel))
        furnace.check(img, w, h, expected)
        img = gpu_ctx.render(cam, V.make_params(w, h, spp=spp, max_depth=1, seed=seed, kernel=kernel))
        furnace.check(img, w, h, 0)     # bounce limit 1: every path that hits returns black


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("case", __import__("furnace").HEAD_ON, ids=lambda c: c[0])
def test_path_head_on_known_answer_on_gpu(gpu_ctx, case, kernel):
    """Second analytic pin of tests/furnace.py (mirror / glass sphere seen head-on), on every PATH kernel."""
    import furnace
    name, kind, albedo, fuzz, ior = case
    sph, mat = furnace.head_on_scene(kind, albedo, fuzz, ior)
    w, h = 513, 385
    cam = V.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60.0, w / h, 0.0, 1.0)
    gpu_ctx.set_scene(sph, mat)
    img = gpu_ctx.render(cam, V.make_params(w, h, spp=4, max_depth=50, seed=5, kernel=kernel))
    furnace.check_head_on(img, kind, albedo)


@pytest.mark.parametrize("kernel", KERNELS)
def test_path_mirror_hall_known_answer_on_gpu(gpu_ctx, kernel):
    """Third analytic pin (tests/mirrors.py), no oracle in the loop: reflection chains between two mirror balls
    and a mirror floor against an independent float64 tracer, on every PATH kernel."""
    import mirrors
    w, h = 360, 240
    sph, mat = mirrors.scene()
    gpu_ctx.set_scene(sph, mat)
    img = gpu_ctx.render(mirrors.camera(w, h), V.make_params(w, h, spp=8, max_depth=mirrors.MAX_DEPTH, seed=3, kernel=kernel))
    mirrors.check(img, w, h)


@pytest.mark.parametrize("kernel", KERNELS)
def test_path_blue_white_hall_known_answer_on_gpu(gpu_ctx, kernel):
    """Fourth analytic pin (tests/furnace.py), no oracle in the loop: blue albedo 1 everywhere, hollow glass
    included -- blue is 255 in every pixel on every PATH kernel."""
    import furnace
    w, h = 120, 80
    sph, mat = furnace.white_hall_scene()
    cam = V.make_camera((0, 0.3, 0.6), (0, 0, -1.6), (0, 1, 0), 55.0, w / h, 0.0, 1.0)
    gpu_ctx.set_scene(sph, mat)
    img = gpu_ctx.render(cam, V.make_params(w, h, spp=16, max_depth=furnace.WHITE_HALL_DEPTH, seed=7, kernel=kernel))
    furnace.check_white_hall(img)


@pytest.mark.parametrize("kernel", KERNELS)
def test_path_glass_ball_known_answer_on_gpu(gpu_ctx, kernel):
    """Fifth analytic pin (tests/glass.py), no oracle in the loop: the expected colour of a glass ball under the gradient
    sky (float64 sum over the reflect/refract tree) against 2048-spp renderings of every PATH kernel, within 3 bytes."""
    import glass
    w, h = 96, 64
    sph, mat = glass.scene()
    gpu_ctx.set_scene(sph, mat)
    img = gpu_ctx.render(glass.camera(w, h), V.make_params(w, h, spp=2048, max_depth=50, seed=3, kernel=kernel))
    assert glass.check(img, w, h, tol=3) <= 3


@pytest.mark.parametrize("kernel", KERNELS)
def test_path_lambertian_ball_known_answer_on_gpu(gpu_ctx, kernel):
    """Sixth analytic pin (tests/lambert.py), no oracle in the loop: albedo * sky(2/3 n.y) for a lone diffuse ball,
    against 2048-spp renderings of every PATH kernel, within 3 bytes."""
    import lambert
    w, h = 96, 64
    sph, mat = lambert.scene()
    gpu_ctx.set_scene(sph, mat)
    img = gpu_ctx.render(lambert.camera(w, h), V.make_params(w, h, spp=2048, max_depth=50, seed=5, kernel=kernel))
    assert lambert.check(img, w, h, tol=3) <= 3


@pytest.mark.parametrize("kernel", KERNELS)
def test_path_fuzzy_metal_ball_known_answer_on_gpu(gpu_ctx, kernel):
    """Seventh analytic pin (tests/fuzzmetal.py), no oracle in the loop: fuzzy metal (fuzz formula + absorption rule)
    against a float64 quadrature over the fuzz vector, every PATH kernel, within 3 bytes."""
    import fuzzmetal
    w, h = 96, 64
    sph, mat = fuzzmetal.scene()
    gpu_ctx.set_scene(sph, mat)
    img = gpu_ctx.render(fuzzmetal.camera(w, h), V.make_params(w, h, spp=2048, max_depth=50, seed=5, kernel=kernel))
    assert fuzzmetal.check(img, w, h, tol=3) <= 3


@pytest.mark.parametrize("kernel", KERNELS)
def test_path_defocused_mirror_ball_known_answer_on_gpu(gpu_ctx, kernel):
    """Eighth analytic pin (tests/defocus.py), no oracle in the loop: the thin-lens camera (lens disk sampled by area,
    origin and direction shifted by the lens offset) against a float64 integral, every PATH kernel, within 3 bytes."""
    import defocus
    w, h = 96, 64
    sph, mat = defocus.scene()
    gpu_ctx.set_scene(sph, mat)
    img = gpu_ctx.render(defocus.camera(w, h), V.make_params(w, h, spp=2048, max_depth=50, seed=5, kernel=kernel))
    assert defocus.check(img, w, h, tol=3) <= 3


def test_ch05_stretched_reference_jpeg_on_gpu(gpu_ctx):
    """RTCHAP05/RTCHAP05/1728.jpg straight against the HIP kernel (no oracle): UBO {1024, 1024, 2, 2, 1}, flipped
    and sampled at the 800x600 window's pixel centres; same bounds as the oracle's test of this fixture."""
    st = json.load(open(os.path.join(GOLD, "ref_jpeg_stats.json")))["RTCHAP05/RTCHAP05/1728.jpg"]
    img = gpu_ctx.render_ubo(V.RtUbo5(*st["ubo"]), V.RT_MODE_CH05)[::-1]
    cw, ch = st["compute_image"]
    w, h = st["size"]
    shown = img[((np.arange(h) + 0.5) / h * ch).astype(int)][:, ((np.arange(w) + 0.5) / w * cw).astype(int)]
    hit = (shown[..., 0] == 255) & (shown[..., 1] == 0)
    ys, xs = np.nonzero(hit)
    for got, want in zip((xs.min(), xs.max(), ys.min(), ys.max()), st["red_bbox_x"] + st["red_bbox_y"]):
        assert abs(int(got) - want) <= st["bbox_tolerance"]
    assert abs(int(hit.sum()) - st["red_count"]) <= st["count_tolerance"] * st["red_count"]
    for key, (y, x) in {"top_left": (0, 0), "bottom_right": (h - 1, w - 1), "centre": (h // 2, w // 2)}.items():
        assert np.abs(shown[y, x, :3].astype(int) - np.array(st[key])).max() <= st["jpeg_tolerance"], key


def test_frames_in_flight_on_three_contexts(gpu_ctx, oracle):
    """Three contexts with different scenes and cameras render concurrently on their own streams (what
    bench.py does per rank at N > 1, and what the reference's per-swapchain-image command buffers do,
    RTCHAP06/main.cpp:94-98): every frame equals the one the context renders alone."""
    torch = pytest.importorskip("torch")
    w, h = 160, 90
    jobs = []
    for k, (name, spp) in enumerate([("cover11", 6), ("three", 9), ("cover5", 3)]):
        sph, mat, cam = _case(oracle, name, w, h)
        prm = V.make_params(w, h, spp=spp, max_depth=12, seed=20 + k)
        gpu_ctx.set_scene(sph, mat)
        alone = gpu_ctx.render(cam, prm)
        ctx = V.Context(0)
        ctx.set_scene(sph, mat)
        jobs.append((ctx, cam, prm, alone, torch.cuda.Stream(),
                     [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(4)]))
    torch.cuda.synchronize()
    for rnd in range(4):            # 12 frames in flight over three streams
        for ctx, cam, prm, _, ts, bufs in jobs:
            ctx.render_device(cam, prm, bufs[rnd].data_ptr(), w * 4, ts.cuda_stream)
    torch.cuda.synchronize()
    for ctx, _, _, alone, _, bufs in jobs:
        for b in bufs:
            assert np.array_equal(b.cpu().numpy().view(np.uint8).reshape(h, w, 4), alone)
        ctx.close()


def test_one_context_two_streams_is_serialised(gpu_ctx, oracle):
    """A context has one frame in flight: its pixel queues, accumulators and chunk order are its own.  Renders
    issued back to back on two different streams are ordered on the device (hipStreamWaitEvent on the end of
    the previous one) instead of sharing the queues: every frame is the oracle's."""
    torch = pytest.importorskip("torch")
    w, h = 200, 120
    sph, mat, cam = _case(oracle, "cover11", w, h)
    ctx = V.Context(0)
    ctx.set_scene(sph, mat)
    prms = [V.make_params(w, h, spp=6, max_depth=50, seed=31), V.make_params(w, h, spp=4, max_depth=50, seed=32)]
    wants = [oracle.render(sph, mat, cam, p)[0] for p in prms]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(6)]
    torch.cuda.synchronize()
    for k in range(6):   # alternate streams and frames without any host sync in between
        ctx.render_device(cam, prms[k % 2], bufs[k].data_ptr(), w * 4, streams[k % 2].cuda_stream)
    torch.cuda.synchronize()
    for k in range(6):
        got = bufs[k].cpu().numpy().view(np.uint8).reshape(h, w, 4)
        assert np.array_equal(got, wants[k % 2]), k
    ctx.close()


@pytest.mark.parametrize("kernel", [V.KERNEL_PERSISTENT, V.KERNEL_CLUSTERED, V.KERNEL_CLUSTERED_PASS])
def test_cost_ordered_dequeue_never_changes_the_frame(gpu_ctx, oracle, kernel):
    """The second and later frames of one shape hand their pixels out in the order of the previous frame's
    per-chunk cost (rtiow_device.h: chunk_order); a changed size, tile or scene starts over.  Scheduling only:
    every frame is the oracle's, and so is a frame of another shape in between."""
    w, h = 257, 131          # ragged last chunk
    sph, mat, cam = _case(oracle, "cover11", w, h)
    ctx = V.Context(0)
    ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=5, max_depth=50, seed=8, kernel=kernel)
    want, segs = oracle.render(sph, mat, cam, prm)
    for _ in range(4):
        assert np.array_equal(ctx.render(cam, prm), want)
        assert ctx.stats().segments == segs
    tile = V.make_params(w, h, spp=5, max_depth=50, seed=8, kernel=kernel, row_block=3, tile_rank=1, tile_count=4)
    want_t, _ = oracle.render(sph, mat, cam, tile)
    for _ in range(3):
        assert np.array_equal(ctx.render(cam, tile), want_t)
    assert np.array_equal(ctx.render(cam, prm), want)
    ctx.close()


def test_ch_via_rtrender_keeps_the_callers_height(gpu_ctx, oracle):
    """main.cpp:103-106 computes the image height as float(w)/(float(w)/float(h)) and truncates it: for ~5 % of
    sizes that is h - 1.  rtRenderUbo is faithful to that; rtRender, whose caller gave an integer height,
    renders all h rows (the UBO floats are kept for the u/v arithmetic)."""
    for w, h in ((2, 7), (2, 13), (400, 225)):
        ubo = oracle.ubo_from_image(w, h)
        short = oracle.render_ubo(ubo, V.RT_MODE_CH06)
        got = gpu_ctx.render(None, V.make_params(w, h, mode=V.RT_MODE_CH06))
        assert got.shape == (h, w, 4)
        assert np.array_equal(got[:short.shape[0]], short)
        if short.shape[0] < h:
            assert got[h - 1, :, :3].max() > 0      # the last row is rendered, not left untouched
        assert got[..., 3].max() == 0


@pytest.mark.parametrize("kernel", KERNELS)
def test_bounce_limit_one_and_zero(gpu_ctx, oracle, kernel):
    """max_depth 1: every path is one segment (sky or black), identical on all kernels and the oracle.
    max_depth 0 (a zero-initialised RtParams) is refused rather than left to differ between kernels."""
    w, h = 90, 60
    sph, mat, cam = _case(oracle, "cover11", w, h)
    gpu_ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=3, max_depth=1, seed=6, kernel=kernel)
    want, segs = oracle.render(sph, mat, cam, prm)
    assert np.array_equal(gpu_ctx.render(cam, prm), want)
    assert gpu_ctx.stats().segments == segs == w * h * 3
    with pytest.raises(V.RtError) as e:
        gpu_ctx.render(cam, V.make_params(w, h, spp=3, max_depth=0, seed=6, kernel=kernel))
    assert e.value.code == V.RT_ERR_INVALID


def test_progressive_with_an_empty_tile(gpu_ctx, oracle):
    """A rank that owns no rows (more tiles than row blocks) still follows the progressive sequence of its
    peers: its second dispatch is not refused."""
    w, h = 40, 8
    sph, mat, cam = _case(oracle, "three", w, h)
    gpu_ctx.set_scene(sph, mat)
    for off in (0, 2, 4):
        prm = V.make_params(w, h, spp=2, max_depth=10, seed=3, row_block=4, tile_rank=5, tile_count=8,
                            sample_offset=off, accumulate=1)
        assert V.tile_row_count(h, 4, 5, 8) == 0
        assert gpu_ctx.render(cam, prm).shape[0] == 0


def test_error_behaviour(gpu_ctx, oracle):
    lib = V.load_library()
    fresh = V.Context(0)
    cam = V.camera_from_ubo(V.ubo_from_image(8, 8))
    with pytest.raises(V.RtError) as e:
        fresh.render(cam, V.make_params(8, 8, spp=1))      # PATH before rtSetScene
    assert e.value.code == V.RT_ERR_STATE
    sph, mat = V.make_three_sphere_scene()
    fresh.set_scene(sph, mat)
    for bad in (V.make_params(8, 8, spp=0), V.make_params(1, 8), V.make_params(8, 8, mode=99),
                V.make_params(8, 8, quantiser=7), V.make_params(8, 8, tile_rank=2, tile_count=2), V.make_params(8, 8, kernel=5)):
        with pytest.raises(V.RtError) as e:
            fresh.render(cam, bad)
        assert e.value.code == V.RT_ERR_INVALID
    bad_mat = mat.copy()
    bad_mat["kind"][0] = 9
    with pytest.raises(V.RtError):
        fresh.set_scene(sph, bad_mat)
    with pytest.raises(V.RtError):
        fresh.set_scene(sph[:0], mat[:0])
    out = np.zeros((8, 8, 4), np.uint8)
    assert lib.rtRender(fresh._h, C.byref(cam), C.byref(V.make_params(8, 8)), out.ctypes.data, 30, 0, None) == V.RT_ERR_INVALID
    assert b"pitch" in lib.rtGetLastError(fresh._h)
    with pytest.raises(V.RtError) as e:
        V.Context(99)
    assert e.value.code == V.RT_ERR_NO_DEVICE
    fresh.close()


@pytest.mark.gpu
def test_a_camera_that_moves_every_frame(oracle):
    """The frame loop of RTCHAP06/main.cpp:304-360 with a camera that changes every frame: one context, the
    chunk order of frame k made from the costs of frame k - 1 -- another view --, twelve frames of an orbit, then a camera that flies
    away by a factor 1.7 per frame (up the ladder of ranges the cluster boxes are rebuilt for, and past 64 scene diagonals onto the
    flat list) and comes back.  Every frame is the oracle's frame of that view, byte for byte; the re-boxes happen before the frame's
    events are recorded (kernel_ms stays a kernel's time) and are counted in RtSceneStats."""
    import math
    w, h = 160, 100
    sph, mat = V.make_cover_scene(1, 11)
    base = dict(spp=8, max_depth=50, seed=3)
    r0 = math.hypot(13.0, 3.0)
    views = [((r0 * math.cos(0.23 + 0.21 * k), 2.0 + 0.1 * k, r0 * math.sin(0.23 + 0.21 * k)), 20.0) for k in range(12)]
    views += [((13.0 * 1.7 ** k, 2.0 * 1.7 ** k, 3.0 * 1.7 ** k), 20.0 / 1.5 ** k) for k in range(1, 12)]  # out to ~4500 units
    views += [((13.0 * 1.7 ** k, 2.0 * 1.7 ** k, 3.0 * 1.7 ** k), 20.0 / 1.5 ** k) for k in (8, 5, 2, 0)]   # ... and back
    with V.Context(0) as ctx:
        ctx.set_scene(sph, mat)
        assert ctx.scene_stats().cluster_builds == 1
        kernels, builds = [], []
        for frm, fov in views:
            cam = V.make_camera(frm, (0, 0, 0), (0, 1, 0), fov, w / h, 0.05, math.dist(frm, (0, 0, 0)))
            prm = V.make_params(w, h, **base)
            got = ctx.render(cam, prm)
            st = ctx.stats()
            want, segs = oracle.render(sph, mat, cam, prm)
            assert np.array_equal(got, want) and st.segments == segs, (frm, _diff(got, want))
            assert st.kernel_ms < 50.0  # (a kernel's time: the host-side re-box is not inside the events)
            kernels.append(ctx.last_kernel())
            builds.append(ctx.scene_stats().cluster_builds)
        assert builds[11] == 1                      # the orbit never leaves the scene's own range
        assert builds[-1] > builds[11] + 2          # ... the fly-away climbs the ladder and comes down again
        assert V.KERNEL_PERSISTENT in kernels and kernels[-1] == V.KERNEL_CLUSTERED  # flat list beyond 64 diagonals, then back
        assert abs(ctx.scene_stats().range_diags - 2.0) < 1e-9  # back on the scene's own rung


@pytest.mark.gpu
def test_round4_boundary_additions(gpu_ctx, rt):
    """ABI 4: the frame's kernel reports the shader clock it ran at (RtStats.shader_clock_mhz: a plausible MI355X clock after a PATH
    frame, 0 after one of the reference's shaders), rtGetSceneStats says what rtSetScene made of the scene, and the two limits the
    round-4 kernels rely on are refused at the door: max_depth beyond the 19 bits a path's depth is counted in, a tile of 2^29
    pixels or more, and a camera whose rays could be shorter than 2^-30 or longer than 2^40."""
    sph, mat = V.make_cover_scene(1, 11)
    gpu_ctx.set_scene(sph, mat)
    ss = gpu_ctx.scene_stats()
    assert ss.cluster_builds == 1 and ss.n_spheres == len(sph) and ss.n_super == 0 and ss.n_clusters == 32
    assert ss.range_diags == 2.0 and ss.base_range_diags == 2.0 and ss.flat_axis == 1 and 0.0 < ss.scene_build_ms < 1000.0
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    gpu_ctx.render(cam, V.make_params(300, 200, spp=16, max_depth=50, seed=1))
    assert 1000 <= gpu_ctx.stats().shader_clock_mhz <= 2600, gpu_ctx.stats().shader_clock_mhz
    gpu_ctx.render_ubo(V.ubo_from_image(64, 48), V.RT_MODE_CH06)
    assert gpu_ctx.stats().shader_clock_mhz == 0
    with pytest.raises(rt.RtError) as e:
        gpu_ctx.render(cam, V.make_params(64, 48, spp=1, max_depth=1 << 19, seed=1))
    assert e.value.code == V.RT_ERR_INVALID
    gpu_ctx.render(cam, V.make_params(64, 48, spp=1, max_depth=(1 << 19) - 1, seed=1))  # (the largest depth accepted)
    with pytest.raises(rt.RtError) as e:  # 32768 x 16384 = 2^29 pixels: refused before anything is allocated
        gpu_ctx.render_device(cam, V.make_params(32768, 16384, spp=1, max_depth=1, seed=1), 4096, 32768 * 4)
    assert e.value.code == V.RT_ERR_INVALID
    # a camera whose rays could leave [2^-30, 2^40] (the domain of the kernels' short root and reciprocal) or whose image plane is degenerate
    for bad in (V.make_camera((13e-12, 2e-12, 3e-12), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.0, 1e-11),   # a scene of picometres
                V.make_camera((13e13, 2e13, 3e13), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.0, 1e14)):       # ... of light-days
        with pytest.raises(rt.RtError) as e:
            gpu_ctx.render(bad, V.make_params(64, 48, spp=1, max_depth=1, seed=1))
        assert e.value.code == V.RT_ERR_INVALID
    flat = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    for k in range(3):
        flat.vertical[k] = flat.horizontal[k]
    with pytest.raises(rt.RtError) as e:
        gpu_ctx.render(flat, V.make_params(64, 48, spp=1, max_depth=1, seed=1))
    assert e.value.code == V.RT_ERR_INVALID


@pytest.mark.parametrize("kernel", KERNELS)
def test_cameras_at_the_edges_of_the_accepted_range_equal_the_oracle(gpu_ctx, oracle, kernel):
    """rtRender's precondition on the camera (rtCameraIsRenderable: every ray from the lens to the image plane between 2^-30 and 2^40
    long) is what lets the kernels normalise a camera ray by their six-instruction square root and three-instruction reciprocal; the only
    GPU evidence for those was the scan of one float at a time (rtSelfTestUnaryScan).  Here whole frames are rendered by cameras that sit
    AT the accepted edges -- the image plane as close to and as far from the lens as the check lets through, found by bisection -- and by
    one whose lens is four times its focus distance, and compared with the oracle bit for bit."""
    w, h = 64, 40

    def cam_of(focus, aperture=0.0, frm=(0.0, 0.0, 0.0)):  # (at the origin: the float error of a ray is a few ulps of its LONGEST term)
        return V.make_camera(frm, (0.0, 0.0, -1.0), (0, 1, 0), 40.0, w / h, aperture, focus)

    def edge(lo, hi, ok_at_lo):  # the focus distance nearest the refused side that is still accepted (bisection over the exponent)
        for _ in range(40):
            mid = (lo * hi) ** 0.5
            if V.camera_is_renderable(cam_of(mid)) == ok_at_lo:
                lo = mid
            else:
                hi = mid
        return lo if ok_at_lo else hi
    nearest = edge(2.0 ** -33, 1.0, False)      # refused at 2^-33, accepted at 1
    farthest = edge(1.0, 2.0 ** 42, True)       # accepted at 1, refused at 2^42
    assert 2.0 ** -30.5 < nearest < 2.0 ** -29 and 2.0 ** 37 < farthest < 2.0 ** 40.5, (nearest, farthest)  # (farthest: the frame's corner ray, with u, v up to 2, is the 2^40 one)
    assert not V.camera_is_renderable(cam_of(nearest * 0.98)) and not V.camera_is_renderable(cam_of(farthest * 1.02))
    sph, mat = V.make_three_sphere_scene(True)
    gpu_ctx.set_scene(sph, mat)
    cases = [cam_of(nearest), cam_of(nearest * 1.37), cam_of(farthest), cam_of(farthest * 0.61), cam_of(1.0, aperture=4.0),
             cam_of(2.0 ** -20, aperture=2.0 ** -19)]
    for cam in cases:
        assert V.camera_is_renderable(cam)
        prm = V.make_params(w, h, spp=8, max_depth=50, seed=5, kernel=kernel)
        want, segs = oracle.render(sph, mat, cam, prm)
        got = gpu_ctx.render(cam, prm)
        assert np.array_equal(got, want), _diff(got, want)
        assert gpu_ctx.stats().segments == segs
        assert len(np.unique(want[..., :3].reshape(-1, 3), axis=0)) > 50   # (a picture, not a constant)
