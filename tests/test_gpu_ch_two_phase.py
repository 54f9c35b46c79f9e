"""The two-phase pixels of ch_kernel_rows (csrc/rtiow_kernels.hip, "Two-phase pixels"): one-ulp hardware approximations plus a proof
that the stored bytes cannot depend on the difference, the shaders' exact arithmetic where the proof does not reach.  Each premise
of that proof is measured here on the GPU, piece by piece (rtSelfTestArith ops 10-16, rtSelfTestChSkySteps), and whole frames are
compared with the exact kernels' (the knobs build with RTIOW_DEBUG_CH_LEAN) and the oracle's."""
import importlib.util
import os

import numpy as np
import pytest

import vulkan_rtiow_amd as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASE2 = np.uint32(1 << 31)
f32 = np.float32


@pytest.fixture(scope="module")
def gen():
    """tools/gen_ch_sky_table.py (the generator of csrc/rtiow_ch_sky_table.h) and what it builds: table, zones, steps."""
    spec = importlib.util.spec_from_file_location("gen_ch_sky_table", os.path.join(ROOT, "tools", "gen_ch_sky_table.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.table, mod.zones, mod.steps = mod.build()
    return mod


def _bits(x):
    return np.asarray(x, f32).view(np.uint32)


def _arith(ctx, op, a, b=None, c=None):
    a = np.asarray(a, f32)
    b = np.zeros_like(a) if b is None else np.asarray(b, f32)
    c = np.zeros_like(a) if c is None else np.asarray(c, f32)
    return ctx.selftest_arith(op, a, b, c)


def test_v_rsq_f32_is_within_one_ulp(gpu_ctx):
    """Premise of both guard bands: v_rsq_f32(x) = x^-1/2 (1 + e), |e| <= 2^-23.  Every float of two adjacent binades (the instruction
    works on the mantissa and the exponent's parity), and a million floats over the range the lean kernels can see (2^-42 .. 2^42)."""
    rng = np.random.default_rng(5)
    everything = np.arange(0x3E800000, 0x3F800000, dtype=np.uint32).view(f32)  # [0.25, 1)
    wide = (rng.uniform(1.0, 2.0, 1 << 20) * 2.0 ** rng.integers(-42, 42, 1 << 20)).astype(f32)
    for x in (everything, wide):
        got = _arith(gpu_ctx, 10, x).astype(np.float64)
        want = 1.0 / np.sqrt(x.astype(np.float64))
        rel = np.abs(got - want) / want
        assert rel.max() <= 2.0 ** -23, rel.max() * 2.0 ** 23


def test_sky_steps_on_the_gpu_are_the_generators(gpu_ctx, gen):
    """The table rests on the list of floats where the sky colour changes.  The generator derives it on the host from its own float32
    restatement of the shader; here the device code (ch_sky_colour<true>) is evaluated on EVERY float of [-1.0001, 1.0001] -- 2.1
    billion -- and must change colour at the same floats, between the same colours."""
    got = gpu_ctx.selftest_ch_sky_steps(-1.0001, 1.0001)
    want = gen.steps
    assert len(got) == len(want) == 280
    assert np.array_equal(got["unit_y"], np.array([s[0] for s in want], f32))
    assert np.array_equal(got["before"], np.array([s[1] for s in want], np.uint32))
    assert np.array_equal(got["after"], np.array([s[2] for s in want], np.uint32))
    # ... and the colours at the ends of the range, so that "same steps" is "same function"
    ends = _bits(_arith(gpu_ctx, 13, np.array([-1.0001, 1.0001], f32)))
    assert list(ends) == [want[0][1], want[-1][2]]


def test_sky_table_on_the_gpu_is_the_generators(gpu_ctx, gen):
    """Phase 1 alone (op 16: bucket, two comparisons): the library's table against the generator's on floats all over the range, at every
    bucket edge and around every zone; flagged exactly where the generator flags, the same colour elsewhere -- and that colour is the
    exact one for every float within the guard band (the generator's verify(), on these very samples)."""
    lo = np.array([t[0] for t in gen.table], f32)
    hi = np.array([t[1] for t in gen.table], f32)
    below = np.array([t[2] for t in gen.table], np.uint32)
    above = np.array([t[3] for t in gen.table], np.uint32)
    rng = np.random.default_rng(6)
    ys = [rng.uniform(-1.0001, 1.0001, 1 << 21).astype(f32)]
    for z in gen.zones:
        for edge in (z[0], z[1]):
            ys.append(gen.key_to_float(gen.float_to_key(edge) + np.arange(-2000, 2001)))
            ys.append((float(edge) + np.linspace(-4 * gen.GUARD, 4 * gen.GUARD, 1001)).astype(f32))
    for k in range(gen.N_BUCKETS):
        ys.append(gen.key_to_float(gen.float_to_key(f32((k - gen.SCALE) / gen.SCALE)) + np.arange(-64, 65)))
    ys = np.concatenate(ys)
    ys = ys[ys <= f32(1.0001)]  # (the kernel's unit_y' is 1 + 2^-21 at most: bucket 512 is the last)
    got = _bits(_arith(gpu_ctx, 16, ys))
    i = np.minimum(gen.bucket_of(ys), gen.N_BUCKETS - 1)
    flagged = ~((ys < lo[i]) | (ys > hi[i]))
    want = np.where(flagged, PHASE2, np.where(ys < lo[i], below[i], above[i]))
    assert np.array_equal(got, want)
    assert 1000 < flagged.sum() < ys.size // 2
    ok = ~flagged
    for d in (-gen.GUARD, 0.0, gen.GUARD):
        assert np.array_equal(gen.F((ys.astype(np.float64) + d).astype(f32))[ok], got[ok])


def test_sky_two_phase_equals_the_exact_sky(gpu_ctx, gen):
    """(dy, dot(dir, dir)) pairs: random ones, and ones aimed at the steps -- unit_y within a few guard bands of every zone, where the first
    phase must hand over: same 24 bits as the exact arithmetic everywhere, both phases taken, and |dy * rsq(qa) - unit_y| within the bound
    the guard band is sized for."""
    rng = np.random.default_rng(7)
    n = 1 << 20
    qa = (rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-30, 30, n)).astype(f32)
    target = rng.uniform(-1.0, 1.0, n)
    edges = np.array([float(e) for z in gen.zones for e in (z[0], z[1])])
    aimed = rng.choice(edges, n // 2) + rng.uniform(-3 * gen.GUARD, 3 * gen.GUARD, n // 2)
    target[: n // 2] = np.clip(aimed, -1.0, 1.0)
    dy = (target * np.sqrt(qa.astype(np.float64))).astype(f32)
    keep = np.abs(dy.astype(np.float64)) <= np.sqrt(qa.astype(np.float64))  # (qa contains dy * dy in the kernel)
    dy, qa = dy[keep], qa[keep]
    two = _bits(_arith(gpu_ctx, 11, dy, qa))
    exact = _bits(_arith(gpu_ctx, 12, dy, qa))
    assert np.array_equal(two & ~PHASE2, exact)
    second = (two & PHASE2) != 0
    assert second.sum() > 10000 and (~second).sum() > 10000
    # the distance the guard band covers: phase 1's unit_y against the shader's
    y1 = _arith(gpu_ctx, 3, dy, _arith(gpu_ctx, 10, qa)).astype(np.float64)                # dy * v_rsq_f32(qa)
    unit_y = _arith(gpu_ctx, 9, dy, _arith(gpu_ctx, 8, qa)).astype(np.float64)            # RN(dy / RN(sqrt(qa)))
    assert np.abs(y1 - unit_y).max() <= 5.01 * 2.0 ** -24 < 0.63 * gen.GUARD


def test_normal_two_phase_equals_the_exact_normal_colour(gpu_ctx):
    """v = rayAt(t) - centre of any length and direction, and directions aimed at the quantiser's steps (127.5 n + 128 within 2^-12 of an
    integer): same 24 bits, both phases taken, the second by no more than a few times the 3 * 2 * 2^-13 of the channels' guard bands."""
    rng = np.random.default_rng(8)
    n = 1 << 21
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    # aim one component of the second half at a byte boundary: n_c = (k + eps - 128) / 127.5
    k = rng.integers(1, 255, n // 2)
    nc = (k + rng.uniform(-2.0 ** -12, 2.0 ** -12, n // 2) - 128.0) / 127.5
    rest = d[: n // 2, 1:] / np.linalg.norm(d[: n // 2, 1:], axis=1)[:, None] * np.sqrt(1.0 - nc ** 2)[:, None]
    d[: n // 2] = np.roll(np.column_stack([nc, rest]), rng.integers(0, 3), axis=1)
    length = np.where(rng.random(n) < 0.5, 0.5, 2.0 ** rng.uniform(-10, 10, n))
    v = (d * length[:, None]).astype(f32)
    two = _bits(_arith(gpu_ctx, 14, v[:, 0], v[:, 1], v[:, 2]))
    exact = _bits(_arith(gpu_ctx, 15, v[:, 0], v[:, 1], v[:, 2]))
    assert np.array_equal(two & ~PHASE2, exact)
    second = (two & PHASE2) != 0
    assert second[: n // 2].mean() > 0.02            # the aimed ones: 2^-13 of 2^-12 either side, and the other two channels
    assert 1e-4 < second[n // 2:].mean() < 3e-3      # the random ones: ~ 3 channels x 2 x 2^-13 = 7e-4
    # the distance the guard band D = 2^-13 covers, operation by operation on the GPU: the shader's y = (0.5 (v_c / sqrt(d) + 1)) 255 + 0.5
    # against phase 1's fma(v_c * v_rsq_f32(d), 127.5, 128) -- bound 7.7e-5, D = 1.22e-4
    zero = np.zeros(n, f32)
    one = np.ones(n, f32)
    mul = lambda a, b: _arith(gpu_ctx, 3, a, b)
    add = lambda a, b: _arith(gpu_ctx, 4, a, b)
    d = add(add(mul(v[:, 0], v[:, 0]), mul(v[:, 1], v[:, 1])), mul(v[:, 2], v[:, 2]))  # gdot(v, v)
    l = _arith(gpu_ctx, 8, d)
    inv = _arith(gpu_ctx, 10, d)
    worst = 0.0
    for c in range(3):
        nrm = _arith(gpu_ctx, 9, v[:, c], l)
        y = add(mul(mul(np.full(n, 0.5, f32), add(nrm, one)), np.full(n, 255.0, f32)), np.full(n, 0.5, f32))
        y1 = _arith(gpu_ctx, 0, mul(v[:, c], inv), np.full(n, 127.5, f32), np.full(n, 128.0, f32))
        worst = max(worst, float(np.abs(y.astype(np.float64) - y1.astype(np.float64)).max()))
    assert worst <= 7.7e-5 < 2.0 ** -13, worst


@pytest.mark.parametrize("mode", [V.RT_MODE_CH05, V.RT_MODE_CH06])
def test_two_phase_frames_equal_the_exact_kernels_and_the_oracle(gpu_ctx, knobs_ctx, oracle, mode, monkeypatch):
    """Whole frames.  Small ones against the oracle (random moderate UBOs: the sphere anywhere from a dot to larger than the frame); large
    ones -- up to 8192 x 8192, where tens of thousands of pixels take the second phase -- against the exact lean kernel
    (RTIOW_DEBUG_CH_LEAN in the knobs build), byte for byte."""
    rng = np.random.default_rng(40 + mode)
    for _ in range(16):
        w, h = int(rng.integers(2, 900)), int(rng.integers(2, 500))
        u = V.ubo_from_image(w, h)
        u.viewportWidth = float(f32(10.0 ** rng.uniform(-2, 2)))
        u.viewportHeight = float(f32(10.0 ** rng.uniform(-2, 2)))
        u.focalLength = float(f32(10.0 ** rng.uniform(-2, 2)))
        got, want = gpu_ctx.render_ubo(u, mode), oracle.render_ubo(u, mode)
        assert np.array_equal(got, want), (w, h, u.viewportWidth, u.viewportHeight, u.focalLength)
    sizes = [(4096, 4096, None), (8192, 8192, None), (16384, 2048, None), (5000, 3001, (3.0, 1.7, 0.6)), (8192, 4096, (0.9, 0.9, 2.5))]
    for w, h, cam in sizes:
        u = V.ubo_from_image(w, h)
        if cam:
            u.viewportWidth, u.viewportHeight, u.focalLength = cam
        two = gpu_ctx.render_ubo(u, mode)
        monkeypatch.setenv("RTIOW_DEBUG_CH_LEAN", "1")
        exact = knobs_ctx.render_ubo(u, mode)
        monkeypatch.delenv("RTIOW_DEBUG_CH_LEAN", raising=False)
        assert np.array_equal(two, exact), (w, h, cam, int((two != exact).any(axis=2).sum()))
        assert np.array_equal(knobs_ctx.render_ubo(u, mode), two)  # (the knobs build without the knob: the same kernel)
