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

    # the whole frame against the oracle, all 800 rows (a few seconds on the GPU box's host cores)
    want, segs = oracle.render(sph, mat, cam, base)
    assert segs == st.segments
    assert np.array_equal(full, want), int((full != want).any(axis=2).sum())


def test_config4_cover_500spp_all_eight_tiles_vs_oracle(gpu_ctx, oracle):
    """BASELINE config 4 (500 spp, 8 GPUs): every rank's tile of the 8-way block-cyclic split is rendered (one
    after the other on this box's one GPU) and 64 rows of the frame -- every 25th row from rows 0 and 12, which
    touches all eight tiles eight times -- are compared with the oracle's same rows; the tiles' segment counts add
    up to the single-GPU frame's."""
    w, h, spp = 1200, 800, 500
    sph, mat, cam = _cover(w, h)
    gpu_ctx.set_scene(sph, mat)
    tiles = []
    total = 0
    for rank in range(8):
        prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=rank, tile_count=8)
        tiles.append(gpu_ctx.render(cam, prm))
        total += gpu_ctx.stats().segments
        assert tiles[-1].shape[0] == V.tile_row_count(h, 4, rank, 8) == 100
    owners = set()
    for first in (0, 12):      # oracle rows first, first + 25, ... (row_block 1, 25 tiles: rows r with r % 25 == first)
        o = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=1, tile_rank=first, tile_count=25)
        want, _ = oracle.render(sph, mat, cam, o)
        assert want.shape[0] == 32
        for k in range(32):
            row = first + 25 * k
            rank = (row // 4) % 8
            lr = (row // 32) * 4 + row % 4
            assert V.tile_global_row(lr, 4, rank, 8) == row
            assert np.array_equal(tiles[rank][lr], want[k]), row
            owners.add(rank)
    assert owners == set(range(8))
    full = gpu_ctx.render(cam, V.make_params(w, h, spp=spp, max_depth=50, seed=1))
    assert gpu_ctx.stats().segments == total
    for rank in range(8):
        rows = [V.tile_global_row(lr, 4, rank, 8) for lr in range(100)]
        assert np.array_equal(full[rows], tiles[rank])


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


def test_config5_at_its_stated_1024spp_rows_vs_oracle(gpu_ctx, oracle):
    """BASELINE config 5 as stated: 4099 spheres, 3840x2160, 1024 spp, depth 50 (pools of one pixel, the
    super-cluster level of the list under sustained load).  The whole frame is rendered on the one GPU (about two
    seconds); four of its rows -- sky, horizon, the sphere field, the foreground -- are also rendered as a tile of
    their own, and both are compared with the oracle's same four rows (4 x 3840 x 1024 samples against all
    4099 spheres by brute force: ~10 s of the box's host cores)."""
    w, h, spp = 3840, 2160, 1024
    sph, mat, cam = _cover(w, h, grid_half=32)
    gpu_ctx.set_scene(sph, mat)
    rows = [270, 810, 1350, 1890]
    tile = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=1, tile_rank=270, tile_count=540)
    assert [V.tile_global_row(lr, 1, 270, 540) for lr in range(4)] == rows
    want, segs = oracle.render(sph, mat, cam, tile)
    part = gpu_ctx.render(cam, tile)
    assert gpu_ctx.stats().segments == segs
    assert np.array_equal(part, want)
    full = gpu_ctx.render(cam, V.make_params(w, h, spp=spp, max_depth=50, seed=1))
    st = gpu_ctx.stats()
    assert st.paths == w * h * spp and gpu_ctx.last_kernel() == V.KERNEL_CLUSTERED
    assert np.array_equal(full[rows], want)


def test_full_size_progressive_and_pitch_with_whole_chunk_pools(gpu_ctx, oracle):
    """At full size a wave takes whole 32-pixel chunks and writes each 128-byte line at once (rtiow_kernels.hip, line
    buffers).  Here that path runs with everything that touches its addressing: progressive accumulation (two dispatches
    of 3 + 5 spp == one of 8), a destination pitch wider than the row (device buffer), 1200 not being a multiple of 32
    (chunks that run over the end of a row) and both persistent kernels; every 40th row against the oracle at 8 spp."""
    torch = pytest.importorskip("torch")
    w, h = 1200, 800
    sph, mat, cam = _cover(w, h)
    base = V.make_params(w, h, spp=8, max_depth=50, seed=3)
    want, _ = _oracle_rows(oracle, sph, mat, cam, base, 40)
    pitch_px = 1216
    for kernel in (V.KERNEL_CLUSTERED, V.KERNEL_PERSISTENT):
        ctx = V.Context(0)
        ctx.set_scene(sph, mat)
        buf = torch.full((h, pitch_px), 0x7F7F7F7F, dtype=torch.int32, device="cuda:0")
        ts = torch.cuda.Stream()
        ts.wait_stream(torch.cuda.current_stream())
        done = 0
        for spp in (3, 5):
            prm = V.make_params(w, h, spp=spp, max_depth=50, seed=3, kernel=kernel, sample_offset=done, accumulate=1)
            ctx.render_device(cam, prm, buf.data_ptr(), pitch_px * 4, ts.cuda_stream)
            done += spp
        ts.synchronize()
        host = buf.cpu().numpy().view(np.uint8).reshape(h, pitch_px, 4)
        assert np.array_equal(host[::40, :w], want), kernel
        assert (host[:, w:] == 0x7F).all()
        ctx.close()


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
