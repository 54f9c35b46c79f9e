"""GPU (-m gpu): randomized scenes (sizes, scales, materials, hollow spheres, cameras inside and far
outside the scene, apertures) through all three PATH kernels against the oracle, bit for bit."""
import numpy as np
import pytest

import vulkan_rtiow_amd as V

pytestmark = pytest.mark.gpu


def _random_scene(rng, n, scale):
    sph = np.zeros(n, V.SPHERE_DTYPE)
    mat = np.zeros(n, V.MATERIAL_DTYPE)
    sph["cx"] = rng.uniform(-1, 1, n) * scale
    sph["cy"] = rng.uniform(-0.3, 0.3, n) * scale
    sph["cz"] = rng.uniform(-1, 1, n) * scale
    sph["radius"] = rng.uniform(0.02, 0.12, n) * scale * rng.choice([1.0, 1.0, 1.0, 3.0], n)
    kinds = rng.choice([0, 1, 2], n, p=[0.6, 0.25, 0.15])
    mat["kind"] = kinds
    mat["albedo"] = rng.uniform(0.1, 1.0, (n, 3))
    mat["fuzz"] = np.where(kinds == 1, rng.uniform(0, 0.6, n) * (rng.random(n) < 0.7), 0)
    mat["ior"] = np.where(kinds == 2, rng.uniform(1.2, 2.0, n), 0)
    hollow = (kinds == 2) & (rng.random(n) < 0.3)
    sph["radius"] = np.where(hollow, -sph["radius"], sph["radius"])
    if rng.random() < 0.7:      # a ground sphere far larger than everything else
        sph[0] = (0.0, -1000.0 * scale - 0.3 * scale, 0.0, 1000.0 * scale)
        mat[0] = (0, (0.5, 0.5, 0.5), 0.0, 0.0, (0, 0))
    return sph, mat


def fuzz_case(case):
    """Scene, camera and parameters of case `case` of tools/fuzz_kernels.py (wider than the cases below:
    up to 6000 spheres, a few very large ones, scales to 3000, cameras from inside to 40 scene radii)."""
    rng = np.random.default_rng(case)
    n = int(rng.choice([3, 17, 64, 65, 130, 400, 900, 1600, 2500, 4000, 6000]))
    scale = float(rng.choice([0.01, 1.0, 1.0, 50.0, 3000.0]))
    sph, mat = _random_scene(rng, n, scale)
    if rng.random() < 0.3:   # a few more very large spheres
        k = min(int(rng.integers(1, 6)), n - 1)
        if k > 0:
            sph["radius"][1:1 + k] = rng.uniform(5, 60, k) * scale
    w, h = int(rng.integers(16, 160)), int(rng.integers(9, 100))
    dist = float(rng.choice([0.3, 1.0, 2.0, 4.0, 8.0, 40.0])) * scale
    frm = rng.normal(size=3)
    frm = frm / np.linalg.norm(frm) * dist + np.array([0, 0.5 * scale * rng.random(), 0])
    cam = V.make_camera(tuple(frm), (0.0, 0.0, 0.0), (0, 1, 0), float(rng.uniform(15, 110)), w / h,
                        float(rng.choice([0.0, 0.02, 0.3])) * scale, max(dist, 1e-3))
    base = dict(spp=int(rng.integers(1, 9)), max_depth=int(rng.choice([2, 8, 50])), seed=int(rng.integers(0, 2**31)),
                quantiser=int(rng.integers(0, 2)))
    return sph, mat, cam, w, h, base


@pytest.mark.parametrize("case", [100069])
def test_fuzz_regressions(gpu_ctx, oracle, case):
    """Cases tools/fuzz_kernels.py once caught.  100069: scale 3000, camera 24000 away aimed at the origin,
    where the clustered list's padding slots sit: with r^2 = -1 their discriminant hb^2 - (|o|^2 + 1) came
    out positive through the rounding of |d|^2 and a padding slot was taken for a hit."""
    sph, mat, cam, w, h, base = fuzz_case(case)
    want, segs = oracle.render(sph, mat, cam, V.make_params(w, h, **base))
    gpu_ctx.set_scene(sph, mat)
    for kernel in (V.KERNEL_PERSISTENT, V.KERNEL_CLUSTERED, V.KERNEL_CLUSTERED_PASS):
        got = gpu_ctx.render(cam, V.make_params(w, h, kernel=kernel, **base))
        assert int((got != want).any(axis=2).sum()) == 0 and gpu_ctx.stats().segments == segs, (case, kernel)


@pytest.mark.parametrize("case", range(20))
def test_random_scene_all_kernels(gpu_ctx, oracle, case):
    rng = np.random.default_rng(1000 + case)
    n = int(rng.choice([1, 2, 7, 33, 100, 257, 700, 2200]))   # 2200: beyond 96 clusters, super-clusters
    scale = float(rng.choice([0.01, 1.0, 1.0, 50.0, 3000.0]))
    sph, mat = _random_scene(rng, n, scale)
    w, h = int(rng.integers(9, 90)), int(rng.integers(5, 60))
    dist = float(rng.choice([0.5, 2.0, 2.0, 6.0, 40.0, 1500.0])) * scale   # inside, near, far, absurdly far
    frm = rng.normal(size=3)
    frm = frm / np.linalg.norm(frm) * dist + np.array([0, 0.5 * scale * rng.random(), 0])
    cam = V.make_camera(tuple(frm), (0.0, 0.0, 0.0), (0, 1, 0), float(rng.uniform(15, 80)), w / h,
                        float(rng.choice([0.0, 0.02, 0.3])) * scale, max(dist, 1e-3))
    spp = int(rng.integers(1, 7))
    depth = int(rng.choice([1, 3, 12, 50]))
    base = dict(spp=spp, max_depth=depth, seed=int(rng.integers(0, 2**31)), quantiser=int(rng.integers(0, 2)))
    want, segs = oracle.render(sph, mat, cam, V.make_params(w, h, **base))
    gpu_ctx.set_scene(sph, mat)
    for kernel in (V.KERNEL_PIXEL, V.KERNEL_PERSISTENT, V.KERNEL_CLUSTERED, V.KERNEL_CLUSTERED_PASS):
        got = gpu_ctx.render(cam, V.make_params(w, h, kernel=kernel, **base))
        st = gpu_ctx.stats()
        bad = int((got != want).any(axis=2).sum())
        assert bad == 0, (case, kernel, n, scale, dist, bad)
        assert st.segments == segs, (case, kernel)


def test_default_kernel_choice(gpu_ctx):
    """RtParams.kernel 0: clustered list from 64 spheres on, flat list for small scenes; an explicit choice
    is honoured.  A camera outside the range the cluster boxes were sized for has them re-boxed with wider
    margins (and re-boxed again when it comes back); only an absurdly distant one gets the flat list."""
    sph, mat = V.make_cover_scene(1, 11)
    gpu_ctx.set_scene(sph, mat)
    near = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    far = V.make_camera((1300, 200, 300), (0, 0, 0), (0, 1, 0), 1.0, 1.5, 0.0, 10.0)
    absurd = V.make_camera((130000, 20000, 30000), (0, 0, 0), (0, 1, 0), 0.01, 1.5, 0.0, 10.0)
    prm = dict(spp=2, max_depth=6, seed=3)
    a = gpu_ctx.render(near, V.make_params(48, 32, **prm))
    assert gpu_ctx.last_kernel() == V.KERNEL_CLUSTERED
    b = gpu_ctx.render(near, V.make_params(48, 32, kernel=V.KERNEL_PERSISTENT, **prm))
    assert gpu_ctx.last_kernel() == V.KERNEL_PERSISTENT and np.array_equal(a, b)
    gpu_ctx.render(near, V.make_params(48, 32, kernel=V.KERNEL_PIXEL, **prm))
    assert gpu_ctx.last_kernel() == V.KERNEL_PIXEL
    f_flat = gpu_ctx.render(far, V.make_params(48, 32, kernel=V.KERNEL_PERSISTENT, **prm))
    f_clus = gpu_ctx.render(far, V.make_params(48, 32, **prm))          # boxes rebuilt for the far camera
    st = gpu_ctx.stats()
    assert gpu_ctx.last_kernel() == V.KERNEL_CLUSTERED and np.array_equal(f_flat, f_clus)
    assert st.sphere_tests < st.segments * len(sph) // 2
    a2 = gpu_ctx.render(near, V.make_params(48, 32, **prm))             # and rebuilt again, tight
    st = gpu_ctx.stats()
    assert gpu_ctx.last_kernel() == V.KERNEL_CLUSTERED and np.array_equal(a, a2)
    assert st.sphere_tests < st.segments * len(sph) // 4
    gpu_ctx.render(absurd, V.make_params(48, 32, **prm))
    assert gpu_ctx.last_kernel() == V.KERNEL_PERSISTENT
    sph3, mat3 = V.make_three_sphere_scene()
    gpu_ctx.set_scene(sph3, mat3)
    gpu_ctx.render(V.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 1.5, 0.0, 1.0), V.make_params(48, 32, **prm))
    assert gpu_ctx.last_kernel() == V.KERNEL_PERSISTENT


@pytest.mark.parametrize("n", [40, 330, 500, 1700])
def test_rays_from_outside_the_box_range(gpu_ctx, oracle, n):
    """A tight knot of spheres on a huge ground, wide camera: most bounces start on the ground far
    outside the range the cluster boxes are inflated for (2 scene diagonals) and must walk every
    cluster; with 330 spheres a wave's pooled work list overflows and each lane walks its own; 500 is about the largest
    scene of the small-scene kernel (32 clusters: the batched sparse trace of a wave with 16 paths fills its work list to the
    last entry); with 1700 the scene has super-clusters and it is their list that overflows."""
    rng = np.random.default_rng(77 + n)
    sph = np.zeros(n + 1, V.SPHERE_DTYPE)
    mat = np.zeros(n + 1, V.MATERIAL_DTYPE)
    sph["cx"][1:] = rng.uniform(-1, 1, n)
    sph["cy"][1:] = rng.uniform(0.1, 2.0, n)
    sph["cz"][1:] = rng.uniform(-1, 1, n)
    sph["radius"][1:] = rng.uniform(0.03, 0.12, n)
    kinds = rng.choice([0, 1, 2], n, p=[0.5, 0.3, 0.2])
    mat["kind"][1:] = kinds
    mat["albedo"][1:] = rng.uniform(0.3, 1.0, (n, 3))
    mat["ior"][1:] = np.where(kinds == 2, 1.5, 0)
    sph[0] = (0.0, -1000.0, 0.0, 1000.0)
    mat[0] = (1, (0.9, 0.9, 0.9), 0.05, 0.0, (0, 0))        # a mirror-like floor sends rays back up to the knot
    cam = V.make_camera((3.5, 1.5, 3.0), (0, 0.5, 0), (0, 1, 0), 100.0, 4 / 3, 0.0, 1.0)
    w, h = 96, 72
    base = dict(spp=6, max_depth=12, seed=5)
    want, segs = oracle.render(sph, mat, cam, V.make_params(w, h, **base))
    gpu_ctx.set_scene(sph, mat)
    for kernel in (V.KERNEL_CLUSTERED, V.KERNEL_CLUSTERED_PASS):
        got = gpu_ctx.render(cam, V.make_params(w, h, kernel=kernel, **base))
        assert gpu_ctx.last_kernel() == V.KERNEL_CLUSTERED
        st = gpu_ctx.stats()
        assert int((got != want).any(axis=2).sum()) == 0 and st.segments == segs
        # ... and such a ray no longer takes every cluster: it tests the boxes enlarged by the margin a ray from that far needs
        # (11 / 31 / 41 / 43 tests per segment; every cluster of every far ray was n / 2 and more)
        assert st.sphere_tests < st.segments * max(16, n // 6), (n, st.sphere_tests / st.segments)


@pytest.mark.parametrize("grid", [34, 36, 38, 39])
def test_scenes_whose_lists_leave_room_for_fewer_waves(gpu_ctx, oracle, grid):
    """Cover scenes of 4625 / 5185 / 5778 / 6086 spheres: the lists take 91 - 118 KB of a CU's LDS and launch_path sizes the workgroup wave by
    wave -- 1024, 896, 704, 640 threads (round 5; groups of 14, 11 and 10 waves are no multiple of the four SIMDs).  The clustered list, with
    and without the primary pass forced, as one dispatch and as two row tiles, against the flat list, and rows of it against the oracle."""
    sph, mat = V.make_cover_scene(1, grid)
    w, h = 96, 48
    cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    base = dict(spp=12, max_depth=12, seed=5)
    gpu_ctx.set_scene(sph, mat)
    flat = gpu_ctx.render(cam, V.make_params(w, h, kernel=V.KERNEL_PERSISTENT, **base))
    segs = gpu_ctx.stats().segments
    # (the oracle, brute force over ~6000 spheres: a quarter of the rows)
    want_rows, _ = oracle.render(sph, mat, cam, V.make_params(w, h, row_block=4, tile_rank=1, tile_count=4, **base))
    assert np.array_equal(flat[np.array([y for y in range(h) if (y // 4) % 4 == 1])], want_rows)
    for kernel in (V.KERNEL_CLUSTERED, V.KERNEL_CLUSTERED_PASS):
        got = gpu_ctx.render(cam, V.make_params(w, h, kernel=kernel, **base))
        assert gpu_ctx.last_kernel() == V.KERNEL_CLUSTERED
        assert np.array_equal(got, flat) and gpu_ctx.stats().segments == segs
        tiles = [gpu_ctx.render(cam, V.make_params(w, h, kernel=kernel, row_block=4, tile_rank=r, tile_count=2, **base)) for r in range(2)]
        for r in range(2):
            assert np.array_equal(tiles[r], flat[np.array([y for y in range(h) if (y // 4) % 2 == r])])


@pytest.mark.parametrize("n", [700, 2500])
def test_many_samples_per_pixel_on_a_large_scene(gpu_ctx, oracle, n):
    """Many samples per pixel on a scene of the large-scene kernels (super-clusters from 40 clusters on: both sizes have them):
    every primary pass of a wave hands out one pixel, pass after pass.  Whole frame against the oracle and against the flat
    list, as one dispatch and as tiles."""
    rng = np.random.default_rng(900 + n)
    sph = np.zeros(n + 1, V.SPHERE_DTYPE)
    mat = np.zeros(n + 1, V.MATERIAL_DTYPE)
    side = int(np.ceil(np.sqrt(n)))
    ij = rng.permutation(side * side)[:n]
    sph["cx"][1:] = (ij % side) - side / 2 + rng.uniform(0.1, 0.7, n)
    sph["cz"][1:] = (ij // side) - side / 2 + rng.uniform(0.1, 0.7, n)
    sph["cy"][1:] = 0.2
    sph["radius"][1:] = 0.2
    kinds = rng.choice([0, 1, 2], n, p=[0.7, 0.2, 0.1])
    mat["kind"][1:] = kinds
    mat["albedo"][1:] = rng.uniform(0.2, 0.9, (n, 3))
    mat["fuzz"][1:] = np.where(kinds == 1, rng.uniform(0, 0.4, n), 0)
    mat["ior"][1:] = np.where(kinds == 2, 1.5, 0)
    sph[0] = (0.0, -1000.0, 0.0, 1000.0)
    mat[0] = (0, (0.5, 0.5, 0.5), 0.0, 0.0, (0, 0))
    w, h = 24, 10
    cam = V.make_camera((9, 2.5, 4), (0, 0, 0), (0, 1, 0), 25.0, w / h, 0.15, 9.0)
    base = dict(spp=400, max_depth=8, seed=11)
    want, segs = oracle.render(sph, mat, cam, V.make_params(w, h, **base))
    gpu_ctx.set_scene(sph, mat)
    flat = gpu_ctx.render(cam, V.make_params(w, h, kernel=V.KERNEL_PERSISTENT, **base))
    assert np.array_equal(flat, want) and gpu_ctx.stats().segments == segs
    for kernel in (V.KERNEL_CLUSTERED, V.KERNEL_CLUSTERED_PASS):
        got = gpu_ctx.render(cam, V.make_params(w, h, kernel=kernel, **base))
        assert gpu_ctx.last_kernel() == V.KERNEL_CLUSTERED
        st = gpu_ctx.stats()
        assert np.array_equal(got, want) and st.segments == segs
        tiles = [gpu_ctx.render(cam, V.make_params(w, h, kernel=kernel, row_block=2, tile_rank=r, tile_count=2, **base)) for r in range(2)]
        rows = [np.array([y for y in range(h) if (y // 2) % 2 == r]) for r in range(2)]
        for r in range(2):
            assert np.array_equal(tiles[r], want[rows[r]])


@pytest.mark.parametrize("case", range(14))
def test_primary_pass_extreme_cameras(gpu_ctx, oracle, case):
    """The cone cull of the primary pass (RtParams.kernel 4 forces it at these sample counts) at the edges of its
    derivation: lenses wider than the focus distance (the cone degenerates: no cull), a field of view of 179 degrees,
    images of two rows / two columns / four pixels (footprints as wide as the view), a camera inside a glass ball,
    inside the sphere field and skimming the ground, a tilted camera, sample counts around the 64 lanes of a pass."""
    sph, mat = V.make_cover_scene(3, 6)
    cams = [
        # (from, at, vfov, aperture, focus, w, h, spp)
        ((13, 2, 3), (0, 0, 0), 20.0, 6.0, 1.0, 40, 24, 9),        # lens radius 3 at focus distance 1
        ((13, 2, 3), (0, 0, 0), 20.0, 2.0, 10.0, 40, 24, 9),       # a very wide lens
        ((3, 0.6, 2), (0, 0.5, 0), 179.0, 0.0, 1.0, 33, 17, 5),    # fish-eye
        ((13, 2, 3), (0, 0, 0), 20.0, 0.1, 10.0, 64, 2, 70),       # two rows (the oracle, like the UBO of main.cpp:103-120, wants at least two)
        ((13, 2, 3), (0, 0, 0), 20.0, 0.1, 10.0, 2, 48, 70),       # two columns
        ((13, 2, 3), (0, 0, 0), 20.0, 0.1, 10.0, 2, 2, 300),       # four pixels
        ((0.0, 1.0, 0.0), (4, 1, 0), 60.0, 0.0, 1.0, 48, 32, 6),   # inside the big glass ball at the origin
        ((0.3, 0.25, 0.4), (4, 0.2, 3), 90.0, 0.05, 2.0, 48, 32, 6),  # between the small spheres
        ((5, 0.01, 5), (0, 0.2, 0), 50.0, 0.02, 5.0, 64, 20, 8),   # skimming the ground
        ((13, 2, 3), (0, 0, 0), 20.0, 0.1, 10.0, 31, 19, 64),      # exactly one pass per pixel
        ((13, 2, 3), (0, 0, 0), 20.0, 0.1, 10.0, 31, 19, 65),
        ((13, 2, 3), (0, 0, 0), 20.0, 0.1, 10.0, 31, 19, 63),
        ((-7, 9, -4), (1, 0, 2), 35.0, 0.3, 11.0, 50, 30, 7),      # from above, tilted (up vector below)
        ((40, 6, 9), (0, 0, 0), 8.0, 0.0, 40.0, 60, 40, 5),        # far and narrow
    ]
    frm, at, fov, ap, focus, w, h, spp = cams[case]
    up = (0.3, 1, 0.2) if case == 12 else (0, 1, 0)
    cam = V.make_camera(frm, at, up, fov, w / h, ap, focus)
    base = dict(spp=spp, max_depth=12, seed=90 + case)
    want, segs = oracle.render(sph, mat, cam, V.make_params(w, h, **base))
    gpu_ctx.set_scene(sph, mat)
    for kernel in (V.KERNEL_CLUSTERED_PASS, V.KERNEL_DEFAULT):
        got = gpu_ctx.render(cam, V.make_params(w, h, kernel=kernel, **base))
        assert gpu_ctx.last_kernel() == V.KERNEL_CLUSTERED
        assert int((got != want).any(axis=2).sum()) == 0 and gpu_ctx.stats().segments == segs, (case, kernel)
    # ... and as row tiles (the cone of a span uses the tile's global rows)
    tiled = np.zeros_like(want)
    for r in range(3):
        rows = [V.tile_global_row(k, 2, r, 3) for k in range(V.tile_row_count(h, 2, r, 3))]
        if rows:
            tiled[rows] = gpu_ctx.render(cam, V.make_params(w, h, kernel=V.KERNEL_CLUSTERED_PASS, row_block=2, tile_rank=r,
                                                            tile_count=3, **base))
    assert np.array_equal(tiled, want), case


@pytest.mark.parametrize("up_axis", [0, 1, 2])
def test_flat_axis_box_test_equals_whole_boxes(oracle, monkeypatch, up_axis):
    """The clustered kernels test a scene's cluster boxes without their flat axis when all boxes span (nearly) one interval
    along it (spheres on a ground plane: rtiow_clusters.cpp, slab_gap_flat).  A layer of spheres lying in each of the
    three coordinate planes in turn (flat axis 0, 1, 2), small (clusters only) and large (super-clusters too), rendered
    with the flat test, with whole boxes (RTIOW_DEBUG_FLAT=0) and -- a scene that is NOT flat -- with the flat test forced
    on (RTIOW_DEBUG_FLAT=1: conservative, only slower): all equal to the oracle's frame."""
    for n, cubic in ((300, False), (2600, False), (500, True)):
        rng = np.random.default_rng(77 + n + up_axis)
        sph, mat = _random_scene(rng, n, 1.0)
        pos = np.stack([sph["cx"], rng.uniform(-0.004, 0.004, n) if not cubic else rng.uniform(-1, 1, n), sph["cz"]], 1)
        sph["radius"] = np.sign(sph["radius"]) * rng.uniform(0.05, 0.06, n)  # (no ground sphere; radii alike, so that a layer is flat)
        pos = np.roll(pos, up_axis - 1, axis=1)  # the layer's normal along up_axis
        sph["cx"], sph["cy"], sph["cz"] = pos[:, 0], pos[:, 1], pos[:, 2]
        w, h = 64, 40
        frm = np.roll(np.array([2.5, 1.2, 1.9]), up_axis - 1)
        cam = V.make_camera(tuple(frm), (0.0, 0.0, 0.0), tuple(np.roll(np.array([0.0, 1.0, 0.0]), up_axis - 1)), 40.0, w / h, 0.05, 3.0)
        base = dict(spp=3, max_depth=12, seed=5)
        want, segs = oracle.render(sph, mat, cam, V.make_params(w, h, **base))
        tests = {}
        for mode in ("0", "1", "2"):  # whole boxes; flat forced; the library's own choice
            monkeypatch.setenv("RTIOW_DEBUG_FLAT", mode)
            with V.Context(0, lib_path=V.api.KNOBS_LIB_PATH) as ctx:  # (the knobs build: the shipped library reads no environment)
                ctx.set_scene(sph, mat)
                for kernel in (V.KERNEL_CLUSTERED, V.KERNEL_CLUSTERED_PASS):
                    got = ctx.render(cam, V.make_params(w, h, kernel=kernel, **base))
                    st = ctx.stats()
                    assert np.array_equal(got, want) and st.segments == segs, (n, cubic, up_axis, mode, kernel)
                    tests[mode, kernel] = st.sphere_tests
        # (which boxes a ray is tested against differs between the modes, the frame must not: asserted above for every mode)
        assert all(v > 0 for v in tests.values())
