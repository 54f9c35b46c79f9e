"""GPU (-m gpu): BASELINE.json's full-size configurations, checked through size-independent
properties (two independent schedulers agree, tiles reassemble, segment counts add up) and against
the oracle on row samples the CPU finishes in seconds."""
import zlib

import numpy as np
import pytest

import vulkan_rtiow_amd as V

pytestmark = pytest.mark.gpu


def _cover(w, h, grid_half=11):
    sph, mat = V.make_cover_scene(1, grid_half)
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    return sph, mat, cam


def _oracle_rows(oracle, sph, mat, cam, base, stride):
    """oracle render of rows 0, stride, 2*stride, ... of the frame described by `base`"""
    prm = V.make_params(base.width, base.height, spp=base.spp, max_depth=base.max_depth, seed=base.seed,
                        quantiser=base.quantiser, row_block=1, tile_rank=0, tile_count=stride)
    img, segs = oracle.render(sph, mat, cam, prm)
    return img, segs


def test_config2_three_spheres_400x225_100spp_full_oracle_parity(gpu_ctx, oracle):
    """BASELINE config 2, every pixel against the oracle."""
    w, h = 400, 225
    sph, mat = V.make_three_sphere_scene(False)
    cam = V.camera_from_ubo(V.ubo_from_image(w, h))
    gpu_ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=100, max_depth=50, seed=1)
    got = gpu_ctx.render(cam, prm)
    st = gpu_ctx.stats()
    want, segs = oracle.render(sph, mat, cam, prm)
    assert np.array_equal(got, want)
    assert st.segments == segs and st.paths == w * h * 100


def test_config3_cover_1200x800_100spp(gpu_ctx, oracle):
    """BASELINE config 3 at full size: persistent kernel == one-lane-per-pixel kernel byte for byte
    (two independent schedules of the same spec), 8 block-cyclic tiles reassemble to the same frame
    with segment counts adding up, and every 100th row equals the oracle."""
    w, h = 1200, 800
    sph, mat, cam = _cover(w, h)
    gpu_ctx.set_scene(sph, mat)
    base = V.make_params(w, h, spp=100, max_depth=50, seed=1)
    full = gpu_ctx.render(cam, base)
    st = gpu_ctx.stats()
    assert st.paths == w * h * 100 and full[..., 3].max() == 0
    v1 = gpu_ctx.render(cam, V.make_params(w, h, spp=100, max_depth=50, seed=1, kernel=V.KERNEL_PIXEL))
    assert gpu_ctx.stats().segments == st.segments
    assert np.array_equal(full, v1)

    frame = np.zeros_like(full)
    total = 0
    for rank in range(8):
        prm = V.make_params(w, h, spp=100, max_depth=50, seed=1, row_block=4, tile_rank=rank, tile_count=8)
        part = gpu_ctx.render(cam, prm)
        total += gpu_ctx.stats().segments
        rows = [V.tile_global_row(lr, 4, rank, 8) for lr in range(part.shape[0])]
        frame[rows] = part
    assert np.array_equal(frame, full) and total == st.segments

    want, _ = _oracle_rows(oracle, sph, mat, cam, base, 100)
    assert np.array_equal(full[::100], want)


def test_config4_cover_500spp_rows_vs_oracle(gpu_ctx, oracle):
    """BASELINE config 4 (500 spp): one rank's tile of an 8-way split against the oracle's same tile
    on four of this tile's rows."""
    w, h = 1200, 800
    sph, mat, cam = _cover(w, h)
    gpu_ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=500, max_depth=50, seed=1, row_block=4, tile_rank=3, tile_count=8)
    part = gpu_ctx.render(cam, prm)
    assert part.shape[0] == V.tile_row_count(h, 4, 3, 8) == 100
    rows = [V.tile_global_row(lr, 4, 3, 8) for lr in range(part.shape[0])]   # 12..15, 44..47, ...
    pick = [rows.index(r) for r in (12, 204, 396, 780)]
    o = V.make_params(w, h, spp=500, max_depth=50, seed=1, row_block=1, tile_rank=12, tile_count=192)
    want, _ = oracle.render(sph, mat, cam, o)             # oracle rows 12, 204, 396, 588, 780
    assert np.array_equal(part[pick], want[[0, 1, 2, 4]])


def test_config5_scene_4096_spheres_3840x2160(gpu_ctx, oracle):
    """BASELINE config 5's scene and resolution (64 KiB sphere list: the 1024-thread / global
    shading-record variant of the kernel), 2 spp on the GPU; every 240th row against the oracle."""
    w, h = 3840, 2160
    sph, mat, cam = _cover(w, h, grid_half=32)
    assert 4000 <= len(sph) <= 4100
    gpu_ctx.set_scene(sph, mat)
    base = V.make_params(w, h, spp=2, max_depth=50, seed=1)
    full = gpu_ctx.render(cam, base)           # default kernel: the clustered list beyond 1024 spheres
    st = gpu_ctx.stats()
    assert st.paths == w * h * 2 and st.sphere_tests < st.segments * len(sph) // 4
    flat = gpu_ctx.render(cam, V.make_params(w, h, spp=2, max_depth=50, seed=1, kernel=V.KERNEL_PERSISTENT))
    sf = gpu_ctx.stats()
    assert np.array_equal(flat, full) and sf.segments == st.segments and sf.sphere_tests == sf.segments * len(sph)
    want, _ = _oracle_rows(oracle, sph, mat, cam, base, 240)
    assert np.array_equal(full[::240], want)
    again = gpu_ctx.render(cam, base)
    assert zlib.crc32(again.tobytes()) == zlib.crc32(full.tobytes())


def test_many_samples_one_pixel_row(gpu_ctx, oracle):
    """spp larger than a pool (pool = 1 pixel): 5000 spp on a 16x2 image, and spp = 1."""
    w, h = 16, 2
    sph, mat = V.make_three_sphere_scene(True)
    cam = V.camera_from_ubo(V.ubo_from_image(w, h))
    gpu_ctx.set_scene(sph, mat)
    for spp in (5000, 1, 33):
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=9)
        got = gpu_ctx.render(cam, prm)
        want, segs = oracle.render(sph, mat, cam, prm)
        assert np.array_equal(got, want), spp
        assert gpu_ctx.stats().segments == segs
