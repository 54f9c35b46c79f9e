"""Out-of-focus mirror ball: an analytic known answer for the THIN-LENS CAMERA that owes nothing to the oracle.

With a perfect mirror a path is a deterministic function of its first ray, and the first ray of the spec's camera
(the book's positionable camera) is a deterministic function of the point in the pixel's footprint and the
point on the lens disk.  A pixel's expected radiance is therefore an integral over footprint x lens disk, which this module
evaluates in float64 (3 x 3 footprint points x 16 x 32 lens points, area-uniform on the disk), with its own restatement
of the camera from the book's formulas.  The ball is well out of focus (focus distance 1.2, ball at 2, aperture 0.3): its
rim is a wide blur whose PROFILE depends on the lens sampling being uniform by AREA (radius sqrt(u)) and on origin and
direction both being shifted by the lens offset.  Renderings at 2048 spp must land within 3 bytes.
"""
import math

import numpy as np

import vulkan_rtiow_amd as V

CENTRE = np.array([0.0, 0.0, -2.0])
RADIUS = 0.5
ALBEDO = np.array([0.8, 0.75, 0.7])
VFOV, APERTURE, FOCUS = 40.0, 0.3, 1.2

_NR, _NA = 16, 32
_r = np.sqrt((np.arange(_NR) + 0.5) / _NR)
_a = (np.arange(_NA) + 0.5) / _NA * 2.0 * np.pi
_LENS = np.stack([np.outer(_r, np.cos(_a)).ravel(), np.outer(_r, np.sin(_a)).ravel()], axis=1)   # equal-area points of the unit disk


def scene():
    sph = np.zeros(1, V.SPHERE_DTYPE)
    mat = np.zeros(1, V.MATERIAL_DTYPE)
    sph[0] = (CENTRE[0], CENTRE[1], CENTRE[2], RADIUS)
    mat[0] = (V.RT_MAT_METAL, tuple(ALBEDO), 0.0, 0.0, (0, 0))
    return sph, mat


def camera(w, h):
    return V.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), VFOV, w / h, APERTURE, FOCUS)


def _sky(d):
    t = 0.5 * (d[:, 1] + 1.0)
    return (1.0 - t)[:, None] * np.ones(3)[None, :] + t[:, None] * np.array([0.5, 0.7, 1.0])[None, :]


def _pixel(i, j, w, h):
    hh = math.tan(math.radians(VFOV) / 2)
    vh, vw = 2 * hh, 2 * hh * w / h
    u, v, wv = np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 0, 1.0])
    horiz, vert = FOCUS * vw * u, FOCUS * vh * v
    llc = -horiz / 2 - vert / 2 - FOCUS * wv
    acc = np.zeros(3)
    for dj in (1 / 6, 0.5, 5 / 6):
        for di in (1 / 6, 0.5, 5 / 6):
            s, t = (i + di) / (w - 1), (j + dj) / (h - 1)
            off = (APERTURE / 2) * (_LENS[:, :1] * u[None, :] + _LENS[:, 1:] * v[None, :])
            o = off
            d = (llc + s * horiz + t * vert)[None, :] - off
            d = d / np.linalg.norm(d, axis=1, keepdims=True)
            oc = o - CENTRE[None, :]
            hb = np.einsum("ij,ij->i", oc, d)
            disc = hb * hb - (np.einsum("ij,ij->i", oc, oc) - RADIUS * RADIUS)
            hit = disc > 0
            tt = -hb - np.sqrt(np.where(hit, disc, 0.0))
            p = o + tt[:, None] * d
            n = (p - CENTRE[None, :]) / RADIUS
            refl = d - 2.0 * np.einsum("ij,ij->i", d, n)[:, None] * n
            refl = refl / np.linalg.norm(refl, axis=1, keepdims=True)
            col = np.where(hit[:, None], ALBEDO[None, :] * _sky(refl), _sky(d))
            acc += col.mean(axis=0)
    return acc / 9.0


def expectations(w, h, step=4):
    out = []
    for j in range(h // 2 - 22, h // 2 + 23, step):
        for i in range(w // 2 - 30, w // 2 + 31, step):
            e = _pixel(i, j, w, h)
            out.append((j, i, np.array([min(255, int(256 * math.sqrt(min(max(x, 0.0), 0.999 ** 2)))) for x in e])))
    return out


def check(img, w, h, tol):
    exp = expectations(w, h)
    worst = 0
    for j, i, want in exp:
        got = img[j, i, :3].astype(int)
        worst = max(worst, int(np.abs(got - want).max()))
        assert np.abs(got - want).max() <= tol, (j, i, got, want)
    return worst
