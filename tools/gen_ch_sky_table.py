#!/usr/bin/env python3
"""Generates vulkan-rtiow_amd/csrc/rtiow_ch_sky_table.h: the sky colour of raytrace06.comp:45-47 / raytrace05.comp:39-40 as a step
function of normalize(dir).y, for the two-phase form of ch_kernel_rows (csrc/rtiow_kernels.hip: "Two-phase pixels").

The sky colour of a pixel is a function F of ONE float, unit_y = normalize(dir).y:
    t = 0.5 * (unit_y + 1);  col = (1 - t) * (1, 1, 1) + t * (0.5, 0.7, 1.0);  rgba8 = trunc(col * 255 + 0.5)
and it reads unit_y only through u = RN(unit_y + 1), a float on the grid 2^-24 below 1 and 2^-23 from 1 to 2.  This script evaluates
H(u) -- the shader's arithmetic in float32, operation by operation as ch_pixel writes it -- on EVERY point of that grid (25 million),
collects every place where the packed colour changes, maps each to the smallest float unit_y whose u reaches it, and merges the
changes that lie within 2^-12 of one another into a zone [first, last] (the green channel's 0.7 * t is inexact: around each of its
steps the byte flips back and forth a few times within a few ulps of u; red and green steps either coincide or lie 0.0052 apart).
Below the first change of a zone F is one colour, from the last change on it is another; inside, the kernel computes.

Table entry for bucket i = trunc(fma(unit_y', 256, 256)), i in 0..512:  { lo, hi, below, above }
    unit_y' <  lo                 ->  colour `below`   (then every float within G of unit_y' lies below the zone's first change)
    unit_y' >  hi                 ->  colour `above`   (... at or above its last change)
    otherwise                     ->  the exact arithmetic (second phase)
with lo = first - G, hi = last + G rounded outwards and G = 2^-21 >= the distance between the fast phase's unit_y' =
dy * v_rsq_f32(qa) and the shader's RN(dy / RN(sqrt(qa))) (2.5 * 2^-23 at most: DESIGN section 4.1).  A bucket without a zone has
lo = hi = +inf.  The script refuses to write a table in which a bucket (widened by 2^-20 for the rounding of the index and by G)
meets two zones, and before it writes anything it proves the table (prove(): for every bucket, the exact interval of floats the kernel's
index sends there, and on each un-flagged part of it no step of F within G and the right colour -- 691 intervals, every float covered)
and spot-checks it against the arithmetic itself (verify(): seven million floats).

    python tools/gen_ch_sky_table.py            # writes the header
    python tools/gen_ch_sky_table.py --check    # regenerates and compares with the committed header (tests/test_host_logic.py)
"""
import os
import sys

import numpy as np

f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "vulkan-rtiow_amd", "csrc", "rtiow_ch_sky_table.h")
N_BUCKETS = 513           # indices 0 .. 512
SCALE = 256.0             # bucket = trunc(fma(y, 256, 256))
GUARD = 2.0 ** -21        # G
MERGE = 2.0 ** -12        # changes closer than this form one zone
EDGE = 2.0 ** -20         # slack on a bucket's edges (the index is a rounded fma)


def H(u):
    """The packed rgba8 sky colour from u = RN(unit_y + 1): ch_pixel's expressions, one float32 rounding per operation."""
    u = u.astype(f32)
    t = f32(0.5) * u
    kk = f32(1.0) - t
    cols = (f32(1.0) * kk + f32(0.5) * t, f32(1.0) * kk + f32(0.7) * t, f32(1.0) * kk + f32(1.0) * t)
    out = np.zeros(u.shape, np.uint32)
    for sh, c in zip((0, 8, 16), cols):
        y = c * f32(255.0) + f32(0.5)  # two roundings (-ffp-contract=off)
        q = np.where(y > 0, y, f32(0)).astype(np.uint32)  # v_cvt_u32_f32: truncation, negatives to 0 (y < 256 here: asserted)
        assert int(q.max()) <= 255
        out |= q << np.uint32(sh)
    return out


def F(y):
    return H(np.asarray(y, f32) + f32(1.0))


def key_to_float(k):
    k = np.asarray(k, np.int64)
    bits = np.where(k < 0, (-k) | 0x80000000, k).astype(np.uint32)
    return bits.view(f32)


def float_to_key(x):
    b = np.asarray(x, f32).view(np.int32).astype(np.int64)
    return np.where(b < 0, -(b & 0x7FFFFFFF), b)


def changes_of_H():
    """[(u_c, colour before, colour from u_c on)] over the whole grid of u, in ascending u."""
    lo = np.arange(-64, 1 << 24, dtype=np.int64).astype(np.float64) * 2.0 ** -24           # [-2^-18, 1)
    hi = 1.0 + np.arange(0, (1 << 23) + 64, dtype=np.int64).astype(np.float64) * 2.0 ** -23  # [1, 2 + 2^-17]
    u = np.unique(np.concatenate([lo, hi]).astype(f32))  # (above 2 the floats are 2^-22 apart: duplicates)
    assert np.all(np.diff(u.astype(np.float64)) > 0)
    c = H(u)
    at = np.nonzero(c[1:] != c[:-1])[0] + 1
    return [(u[i], int(c[i - 1]), int(c[i])) for i in at], int(c[0]), int(c[-1])


def first_unit_y_reaching(u_c):
    """The smallest float y with RN(y + 1) >= u_c (RN(y + 1) is non-decreasing in y)."""
    klo, khi = int(float_to_key(f32(-1.01))), int(float_to_key(f32(1.01)))
    assert f32(key_to_float(klo) + f32(1.0)) < u_c <= f32(key_to_float(khi) + f32(1.0))
    while khi - klo > 1:  # invariant: RN(val(klo) + 1) < u_c <= RN(val(khi) + 1)
        mid = (klo + khi) // 2
        if f32(key_to_float(mid) + f32(1.0)) >= u_c:
            khi = mid
        else:
            klo = mid
    return f32(key_to_float(khi))


def bucket_of(y):
    """The index the kernel computes: trunc(fma(y, 256, 256)) with the fma rounded to float32 (exact in float64 before that)."""
    v = (np.asarray(y, np.float64) * SCALE + SCALE).astype(f32)
    return np.where(v > 0, v, 0).astype(np.int64)


def build():
    changes, c_first, c_last = changes_of_H()
    steps = [(first_unit_y_reaching(u_c), a, b) for u_c, a, b in changes]
    for (y, a, b), (u_c, _, _) in zip(steps, changes):  # the change really happens at y
        assert int(F(y)) == b and int(F(np.nextafter(y, f32(-2)))) == a
    zones = []  # [first, last, below, above]
    for y, a, b in steps:
        if zones and float(y) - float(zones[-1][1]) < MERGE:
            zones[-1][1], zones[-1][3] = y, b
        else:
            zones.append([y, y, a, b])
    for z0, z1 in zip(zones, zones[1:]):
        assert z0[3] == z1[2] and float(z1[0]) - float(z0[1]) > 4.0 / SCALE * 0.33  # settled colours chain; zones far apart
    table = []
    for i in range(N_BUCKETS):
        b_lo, b_hi = (i - SCALE) / SCALE - EDGE, (i + 1 - SCALE) / SCALE + EDGE
        mine = [z for z in zones if float(z[0]) - GUARD <= b_hi and float(z[1]) + GUARD >= b_lo]
        if len(mine) > 1:
            raise SystemExit(f"bucket {i} meets {len(mine)} zones")
        if mine:
            first, last, below, above = mine[0]
            lo = np.nextafter(f32(float(first) - GUARD), f32(-4))
            hi = np.nextafter(f32(float(last) + GUARD), f32(4))
            assert float(lo) < float(first) - GUARD and float(hi) > float(last) + GUARD
            table.append((lo, hi, below, above))
        else:
            c = int(F(f32((i + 0.5 - SCALE) / SCALE)))
            table.append((f32(np.inf), f32(np.inf), c, c))
    return table, zones, steps


def verify(table, zones, steps):
    """Soundness on the host: (1) every float within G of a change is sent to the second phase by the entry of ITS bucket; (2) what the
    first phase answers for an un-flagged float is F of every float within G of it -- checked on 4096 floats per bucket and around
    every zone, F being constant between zones by construction (the scan of H is exhaustive)."""
    lo = np.array([t[0] for t in table], f32)
    hi = np.array([t[1] for t in table], f32)
    below = np.array([t[2] for t in table], np.uint32)
    above = np.array([t[3] for t in table], np.uint32)

    def phase1(y):
        y = np.asarray(y, f32)
        i = np.minimum(bucket_of(y), N_BUCKETS - 1)
        flagged = ~((y < lo[i]) | (y > hi[i]))
        return np.where(y < lo[i], below[i], above[i]), flagged

    rng = np.random.default_rng(1)
    for y_c, _, _ in steps:
        near = np.concatenate([np.linspace(float(y_c) - GUARD, float(y_c) + GUARD, 4097),
                               key_to_float(float_to_key(y_c) + np.arange(-200, 201))]).astype(f32)
        near = near[np.abs(near.astype(np.float64) - float(y_c)) <= GUARD]
        _, flagged = phase1(near)
        assert flagged.all(), float(y_c)
    ys = np.concatenate([rng.uniform(-1.0001, 1.0001, 1 << 21), np.linspace(-1.0001, 1.0001, 1 << 21)]).astype(f32)
    for z in zones:
        for edge in (z[0], z[1]):
            ys = np.concatenate([ys, key_to_float(float_to_key(edge) + np.arange(-3000, 3001)),
                                 (float(edge) + np.linspace(-4 * GUARD, 4 * GUARD, 2001)).astype(f32)])
    for k in range(N_BUCKETS + 1):
        ys = np.concatenate([ys, key_to_float(float_to_key(f32((k - SCALE) / SCALE)) + np.arange(-64, 65))])
    col, flagged = phase1(ys)
    ok = ~flagged
    for d in (-GUARD, 0.0, GUARD):
        moved = (ys.astype(np.float64) + d).astype(f32)
        assert np.array_equal(F(moved)[ok], col[ok]), d
    return int(flagged.sum()), ys.size


def prove(table, steps):
    """The same claim as verify(), not on samples but for EVERY float: for bucket i, the floats y1 the kernel's index sends there form an
    interval [a_i, b_i] (the index is monotone in y1: found by bisection on the emulated fma + truncation); the first phase answers
    `below` on [a_i, lo) and `above` on (hi, b_i] -- and F, whose every change is in `steps` (the scan of H is exhaustive), must be that
    one colour on the whole of [a_i - G, lo + G) resp. (hi - G, b_i + G]: no step inside, and the colour right."""
    step_y = np.array([float(s[0]) for s in steps])      # F changes AT these floats (first float of the new colour)
    after = [s[2] for s in steps]
    first_colour = steps[0][1]

    def colour_on(x0, x1):
        """F's colour on the real interval [x0, x1] if it is constant there, else None (a step T lies in it when x0 < T <= x1)."""
        k0, k1 = np.searchsorted(step_y, x0, side="right"), np.searchsorted(step_y, x1, side="right")
        if k0 != k1:
            return None
        return after[k0 - 1] if k0 > 0 else first_colour

    klo, khi = int(float_to_key(f32(-1.0 - 2.0 ** -20))), int(float_to_key(f32(1.0 + 2.0 ** -20)))

    def first_key_with_bucket_at_least(i):
        lo, hi = klo - 1, khi + 1  # bucket(lo) < i <= bucket(hi) (sentinels)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if int(bucket_of(key_to_float(mid))) >= i:
                hi = mid
            else:
                lo = mid
        return hi

    starts = [first_key_with_bucket_at_least(i) for i in range(N_BUCKETS + 1)]
    assert starts[0] == klo and starts[N_BUCKETS] == khi + 1, "the kernel's unit_y' (|y1| <= 1 + 2^-21) stays inside the table"
    checked = 0
    for i in range(N_BUCKETS):
        if starts[i + 1] == starts[i]:
            continue
        a_i, b_i = float(key_to_float(starts[i])), float(key_to_float(starts[i + 1] - 1))
        lo, hi, below, above = table[i]
        lo, hi = float(lo), float(hi)
        if a_i < lo:   # floats of the bucket below lo: [a_i, min(b_i, pred(lo))]
            top = min(b_i, float(np.nextafter(f32(lo), f32(-4)))) if np.isfinite(lo) else b_i
            assert colour_on(a_i - GUARD, top + GUARD) == below, (i, "below")
            checked += 1
        if np.isfinite(hi) and b_i > hi:  # floats of the bucket above hi: [max(a_i, succ(hi)), b_i]
            bottom = max(a_i, float(np.nextafter(f32(hi), f32(4))))
            assert colour_on(bottom - GUARD, b_i + GUARD) == above, (i, "above")
            checked += 1
    return checked


def render(table, zones):
    out = ["// GENERATED by tools/gen_ch_sky_table.py -- do not edit.  The sky colour of raytrace06.comp:45-47 as a step function of",
           "// normalize(dir).y: entry i = { lo, hi, colour below lo, colour above hi } for bucket i = trunc(fma(unit_y, 256, 256)); between lo",
           "// and hi (a colour step and the guard band G = 2^-21 either side) the kernel computes.  %d zones, %d buckets." % (len(zones), N_BUCKETS),
           "// tests/test_host_logic.py regenerates it (exhaustive scan of the shader's arithmetic over all 25 million values of unit_y + 1).",
           "#pragma once",
           "#define RTIOW_CH_SKY_BUCKETS %du" % N_BUCKETS,
           "#define RTIOW_CH_SKY_GUARD 0x1p-21f",
           "#define RTIOW_CH_SKY_TABLE_WORDS { \\"]
    for lo, hi, below, above in table:
        out.append("    0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, \\" % (int(f32(lo).view(np.uint32)), int(f32(hi).view(np.uint32)), below, above))
    out.append("}")
    return "\n".join(out) + "\n"


def main():
    table, zones, steps = build()
    flagged, n = verify(table, zones, steps)
    intervals = prove(table, steps)
    text = render(table, zones)
    if "--check" in sys.argv:
        if open(HEADER).read() != text:
            raise SystemExit("rtiow_ch_sky_table.h differs from what tools/gen_ch_sky_table.py generates")
        print(f"ok: {len(steps)} changes in {len(zones)} zones; header up to date; {intervals} un-flagged intervals proven constant within the "
              f"guard band; {flagged} of {n} sample floats take the second phase")
        return
    with open(HEADER, "w") as f:
        f.write(text)
    width = sum(float(z[1]) - float(z[0]) + 2 * GUARD for z in zones)
    print(f"{len(steps)} changes, {len(zones)} zones, second phase for {width / 2:.2e} of unit_y's range; wrote {HEADER}")


if __name__ == "__main__":
    main()
