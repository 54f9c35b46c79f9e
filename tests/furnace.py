"""Blue-channel furnace: an analytic known answer for PATH mode that owes nothing to the oracle.

The sky of raytrace06.comp:45-47 is lerp(white, (0.5, 0.7, 1.0), t): its BLUE component is 1 in every
direction.  A single convex sphere in that sky therefore returns, in blue, exactly its blue albedo for
every camera sample that hits it (the scattered ray leaves a convex body and meets only sky), whatever
the random numbers are: lambertian and mirror metal rho_b, glass 1.  With the book's write_color the
blue byte of a pixel inside the sphere's silhouette is (int)(256 * sqrt(rho_b)) and 255 outside it —
independent of spp, seed, kernel and sampling.  With a bounce limit of 1 the one scatter uses the limit
up and every path that hits returns black: byte 0 inside.  (Albedos are chosen so that 256 sqrt(rho_b) is far from
an integer: single-precision rounding cannot move the byte.)"""
import math

import numpy as np

import vulkan_rtiow_amd as V

CASES = [  # kind, blue albedo, fuzz, ior -> expected blue byte inside the silhouette
    ("lambertian", 0, 0.30, 0.0, 0.0, int(256 * math.sqrt(0.30))),   # 140
    ("mirror", 1, 0.50, 0.0, 0.0, int(256 * math.sqrt(0.50))),       # 181
    ("glass", 2, 1.00, 0.0, 1.5, 255),
]


def scene(kind, rho_b, fuzz, ior):
    sph = np.zeros(1, V.SPHERE_DTYPE)
    mat = np.zeros(1, V.MATERIAL_DTYPE)
    sph[0] = (0.0, 0.0, -1.5, 0.5)
    mat[0] = (kind, (0.9, 0.6, rho_b), fuzz, ior, (0, 0))
    return sph, mat


def check(img, w, h, expected, vfov_deg=60.0):
    """img: [h, w, 4] RGBA8, row 0 = scene bottom.  Sphere centre on the axis at distance 1.5, radius 0.5."""
    blue = img[..., 2].astype(int)
    half_h = math.tan(math.radians(vfov_deg) / 2)
    half_w = half_h * w / h
    ys, xs = np.mgrid[0:h, 0:w]
    px = (xs / (w - 1) * 2 - 1) * half_w     # on the plane z = -1
    py = (ys / (h - 1) * 2 - 1) * half_h
    # angle to the axis against the silhouette's half angle asin(r / d)
    ang = np.arctan(np.hypot(px, py))
    sil = math.asin(0.5 / 1.5)
    pixel = 2 * half_w / (w - 1)
    inside = ang < sil - 2.5 * pixel          # whole pixel footprint (jitter < 1 px) inside the sphere
    outside = ang > sil + 2.5 * pixel
    assert inside.sum() > 50 and outside.sum() > 50
    assert (blue[inside] == expected).all(), np.unique(blue[inside])
    assert (blue[outside] == 255).all(), np.unique(blue[outside])
    # the silhouette itself: a mix of the two, never outside their range
    assert blue.min() >= expected and blue.max() <= 255


# Head-on pixel: the camera ray through the middle of the image meets the sphere along its axis.  A
# mirror sends it straight back, glass lets it straight through (or reflects it straight back): either
# way the path ends in a horizontal direction, where the sky is lerp(white, (0.5, 0.7, 1.0), 0.5) =
# (0.75, 0.85, 1.0).  Expected bytes (int)(256 sqrt(albedo * sky)), +-1 for the sub-pixel jitter.
HEAD_ON = [
    ("mirror", 1, (0.9, 0.6, 0.5), 0.0, 0.0),
    ("glass", 2, (1.0, 1.0, 1.0), 0.0, 1.5),
]


def head_on_expected(kind, albedo):
    sky = (0.75, 0.85, 1.0)
    att = albedo if kind == 1 else (1.0, 1.0, 1.0)
    return [min(255, int(256 * math.sqrt(a * s))) for a, s in zip(att, sky)]


def head_on_scene(kind, albedo, fuzz, ior):
    sph = np.zeros(1, V.SPHERE_DTYPE)
    mat = np.zeros(1, V.MATERIAL_DTYPE)
    sph[0] = (0.0, 0.0, -1.5, 0.5)
    mat[0] = (kind, albedo, fuzz, ior, (0, 0))
    return sph, mat


def check_head_on(img, kind, albedo):
    h, w = img.shape[:2]
    got = img[h // 2, w // 2, :3].astype(int)
    want = np.array(head_on_expected(kind, albedo))
    assert np.abs(got - want).max() <= 1, (got, want)
