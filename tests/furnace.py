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


# Blue-white hall: SEVERAL bodies, every one with blue albedo exactly 1 -- lambertian, mirror, solid glass, a hollow
# glass ball (outer radius 0.5, inner radius -0.4: the negative radius flips the normal) and a
# lambertian floor.  None of these materials absorbs (fuzz 0), the sky's blue is 1 in every direction, so the blue
# radiance of EVERY sample is 1 whatever it bounces off and however often, provided it gets out within the
# bounce limit (400: a path of that length has probability ~0.7^400): the blue byte is 255 in every pixel.
# Any lost path -- a NaN normal, a ray that escapes through the shell's inner surface the wrong way, total
# internal reflection handled as absorption, a hit dropped among several spheres -- darkens a pixel.
def white_hall_scene():
    # (no two surfaces touch or come within 0.05: at a tangent contact the book's t_min = 0.001 lets a ray that
    # starts closer than that to the other surface pass through it, into the floor, where it bounces until the limit)
    rows = [((0.0, -100.56, -1.6), 100.0, V.RT_MAT_LAMBERTIAN, (0.5, 0.6, 1.0), 0.0),
            ((-1.05, 0.0, -1.8), 0.5, V.RT_MAT_LAMBERTIAN, (0.3, 0.5, 1.0), 0.0),
            ((0.0, 0.0, -1.6), 0.5, V.RT_MAT_DIELECTRIC, (1.0, 1.0, 1.0), 1.5),      # hollow glass: outer ...
            ((0.0, 0.0, -1.6), -0.4, V.RT_MAT_DIELECTRIC, (1.0, 1.0, 1.0), 1.5),     # ... and inner surface
            ((1.05, 0.0, -1.8), 0.5, V.RT_MAT_METAL, (0.8, 0.6, 1.0), 0.0),
            ((0.45, -0.3, -0.9), 0.2, V.RT_MAT_DIELECTRIC, (1.0, 1.0, 1.0), 1.5)]
    # gaps: balls (bottom -0.5) to floor (top -0.56); lambertian-shell 0.07; shell-mirror 0.07; small ball to floor 0.06
    sph = np.zeros(len(rows), V.SPHERE_DTYPE)
    mat = np.zeros(len(rows), V.MATERIAL_DTYPE)
    for k, (c, r, kind, alb, ior) in enumerate(rows):
        sph[k] = (c[0], c[1], c[2], r)
        mat[k] = (kind, alb, 0.0, ior, (0, 0))
    return sph, mat


WHITE_HALL_DEPTH = 400


def check_white_hall(img):
    blue = img[..., 2]
    assert (blue == 255).all(), (int((blue != 255).sum()), np.unique(blue)[:8])
    # (and the other channels are not trivially white: the albedos below 1 show)
    assert img[..., 0].min() < 200
