"""Fuzzy-metal ball: an analytic known answer for metal WITH fuzz that owes nothing to the oracle.

The spec's metal (book v4) scatters along normalize(reflect(d, n) + fuzz * v), v uniform on the unit sphere, and absorbs
the path when that direction points into the surface.  Off a lone convex ball a scattered ray always reaches the sky, so
the expected radiance of a surface point is a two-dimensional integral over v,

    albedo * (1 / 4 pi) * Int [ (r + f v) . n > 0 ] * sky( normalize(r + f v) ) dS(v),

which this module evaluates in float64 by the midpoint rule (96 x 192 nodes in z and azimuth), averaged over a 3 x 3 grid
of start points in the pixel's footprint.  It pins the fuzz formula, the absorption rule and -- once more -- the uniformity
of the unit-vector sampler.  Renderings at 2048 spp must land within 3 bytes.
"""
import math

import numpy as np

import vulkan_rtiow_amd as V

CENTRE = np.array([0.0, 0.0, -1.5])
RADIUS = 0.5
ALBEDO = np.array([0.85, 0.7, 0.55])
FUZZ = 0.6
VFOV = 40.0

_NZ, _NPHI = 96, 192
_z = (np.arange(_NZ) + 0.5) / _NZ * 2.0 - 1.0
_phi = (np.arange(_NPHI) + 0.5) / _NPHI * 2.0 * np.pi
_Z, _PHI = np.meshgrid(_z, _phi, indexing="ij")
_R = np.sqrt(1.0 - _Z * _Z)
_VS = np.stack([_R * np.cos(_PHI), _R * np.sin(_PHI), _Z], axis=-1).reshape(-1, 3)   # uniform on the sphere, equal weights


def scene():
    sph = np.zeros(1, V.SPHERE_DTYPE)
    mat = np.zeros(1, V.MATERIAL_DTYPE)
    sph[0] = (CENTRE[0], CENTRE[1], CENTRE[2], RADIUS)
    mat[0] = (V.RT_MAT_METAL, tuple(ALBEDO), FUZZ, 0.0, (0, 0))
    return sph, mat


def camera(w, h):
    return V.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), VFOV, w / h, 0.0, 1.0)


def _point(d):
    hb = -CENTRE @ d
    t = -hb - math.sqrt(hb * hb - (CENTRE @ CENTRE - RADIUS * RADIUS))
    n = (t * d - CENTRE) / RADIUS
    refl = d - 2.0 * (d @ n) * n
    dirs = refl[None, :] + FUZZ * _VS
    keep = dirs @ n > 0.0
    y = dirs[:, 1] / np.linalg.norm(dirs, axis=1)
    ty = 0.5 * (y + 1.0)
    sky = (1.0 - ty)[:, None] * np.ones(3)[None, :] + ty[:, None] * np.array([0.5, 0.7, 1.0])[None, :]
    return ALBEDO * (sky * keep[:, None]).mean(axis=0)


def expectations(w, h, step=7):
    half_h = math.tan(math.radians(VFOV) / 2)
    half_w = half_h * w / h
    sil = math.asin(RADIUS / np.linalg.norm(CENTRE))
    out = []
    for j in range(0, h, step):
        for i in range(0, w, step):
            acc = np.zeros(3)
            inside = True
            for dj in (1 / 6, 0.5, 5 / 6):
                for di in (1 / 6, 0.5, 5 / 6):
                    u, v = (i + di) / (w - 1), (j + dj) / (h - 1)
                    d = np.array([(2 * u - 1) * half_w, (2 * v - 1) * half_h, -1.0])
                    d = d / np.linalg.norm(d)
                    if math.acos(-d[2]) >= 0.85 * sil:
                        inside = False
                        break
                    acc += _point(d)
                if not inside:
                    break
            if inside:
                e = acc / 9.0
                out.append((j, i, np.array([min(255, int(256 * math.sqrt(min(max(x, 0.0), 0.999 ** 2)))) for x in e])))
    return out


def check(img, w, h, tol):
    exp = expectations(w, h)
    assert len(exp) >= 15, len(exp)
    worst = 0
    for j, i, want in exp:
        got = img[j, i, :3].astype(int)
        worst = max(worst, int(np.abs(got - want).max()))
        assert np.abs(got - want).max() <= tol, (j, i, got, want)
    return worst
