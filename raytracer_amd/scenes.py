"""Scene construction on the host, in double, flattened to rt_prim / rt_light records.

default_scene() carries the literals of the reference's SetupScene (src/main.cc:144-207)
through the same type conversions the reference's constructors apply: float radius
(src/objects.h:116), float light / ambient intensity (src/common.h:50-53), float literals
for the materials written with an `f` suffix (src/main.cc:175,188,189), plane normal
normalised once by the constructor (src/objects.h:88).

The synthetic scenes (random spheres, glass spheres) and the camera orbit are not in the
reference; they use a fixed splitmix64 stream so that every run, rank and checker sees
the same scene.
"""
from __future__ import annotations

import math

import numpy as np

from .abi import LIGHT_DTYPE, PRIM_DTYPE, RT_PRIM_PLANE, RT_PRIM_SPHERE

DEFAULT_BGR = (0.58, 0.76, 0.92)  # src/main.cc:221
DEFAULT_FOV = 60.0  # src/main.cc:126
DEFAULT_MAX_DEPTH = 4  # src/main.cc:17
DEFAULT_SIZE = (640, 640)  # src/main.cc:215


def f32(x: float) -> float:
    """The double value of a C float literal / float parameter."""
    return float(np.float32(x))


def make_normal(v) -> tuple[float, float, float]:
    """gmtl::makeNormal: length by a running sum over x,y,z, then three divisions."""
    x, y, z = (float(c) for c in v)
    s = 0.0
    s += x * x
    s += y * y
    s += z * z
    ln = math.sqrt(s)
    if ln != 0.0:
        x, y, z = x / ln, y / ln, z / ln
    return (x, y, z)


def plane(origin, normal, kd, reflectivity=1.0, transparency=0.0, refraction_index=1.0, ambient=0.3):
    """Plane(origin, normal, DiffuseMaterial{...}, Light{_, ambient}) (src/objects.h:84-89)."""
    r = np.zeros((), dtype=PRIM_DTYPE)
    r["type"] = RT_PRIM_PLANE
    r["p"] = origin
    r["q"] = make_normal(normal)
    r["kd"] = kd
    r["reflectivity"], r["transparency"], r["refraction_index"] = reflectivity, transparency, refraction_index
    r["ambient_intensity"] = f32(ambient)
    return r


def sphere(center, radius, kd, reflectivity=1.0, transparency=0.0, refraction_index=1.0, ambient=0.3):
    """Sphere(center, float radius, DiffuseMaterial{...}, Light{_, ambient}) (src/objects.h:116-121)."""
    r = np.zeros((), dtype=PRIM_DTYPE)
    r["type"] = RT_PRIM_SPHERE
    r["p"] = center
    r["q"] = (f32(radius), 0.0, 0.0)
    r["kd"] = kd
    r["reflectivity"], r["transparency"], r["refraction_index"] = reflectivity, transparency, refraction_index
    r["ambient_intensity"] = f32(ambient)
    return r


def point_light(color, intensity, pos):
    """PointLight(color, float intensity, pos) (src/common.h:59-61)."""
    r = np.zeros((), dtype=LIGHT_DTYPE)
    r["pos"], r["color"], r["intensity"] = pos, color, f32(intensity)
    return r


def _stack(records, dtype):
    out = np.zeros(len(records), dtype=dtype)
    for i, r in enumerate(records):
        out[i] = r
    return out


def default_scene():
    """The reference's scene: plane, two spheres, two point lights (src/main.cc:153-206)."""
    half = tuple(c * 0.5 for c in (1.0, 1.0, 1.0))  # Color{1,1,1} * 0.5
    prims = [
        plane((0, -3, 0), (0, 1, 0), half, reflectivity=0.0, transparency=0.0),
        sphere((2, 0.5, 10), 1, (0.5, 0.313, 0.64), reflectivity=0.7, transparency=f32(0.9), refraction_index=1.1),
        sphere((1.5, 0, 13), 1, (0.9, 0.4, 0.298), reflectivity=f32(0.2), transparency=f32(0.0)),
    ]
    lights = [
        point_light((1, 1, 1), 1.0, (0, 4, 7)),
        point_light((1, 1, 1), 0.3, (10, 8, 3)),
    ]
    return _stack(prims, PRIM_DTYPE), _stack(lights, LIGHT_DTYPE)


class SplitMix64:
    """splitmix64: the fixed pseudo-random stream behind the synthetic scenes."""

    MASK = (1 << 64) - 1

    def __init__(self, seed: int):
        self.s = seed & self.MASK

    def next_u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.MASK
        return z ^ (z >> 31)

    def uniform(self, lo: float = 0.0, hi: float = 1.0) -> float:
        u = (self.next_u64() >> 11) * (1.0 / (1 << 53))
        return lo + (hi - lo) * u


def _ring_lights(rng: SplitMix64, count: int, radius: float, cz: float, lo: float = 0.1, hi: float = 0.5):
    lights = []
    for i in range(count):
        ang = 2.0 * math.pi * i / count
        y = rng.uniform(6.0, 12.0)
        inten = rng.uniform(lo, hi)
        lights.append(point_light((1, 1, 1), inten, (radius * math.cos(ang), y, cz + radius * math.sin(ang))))
    return lights


def random_spheres_scene(n_spheres: int = 1024, n_lights: int = 8, seed: int = 0x5EED1024):
    """Ground plane (as in the default scene) at index 0 + n_spheres random spheres."""
    rng = SplitMix64(seed)
    half = (0.5, 0.5, 0.5)
    prims = [plane((0, -3, 0), (0, 1, 0), half, reflectivity=0.0, transparency=0.0)]
    for _ in range(n_spheres):
        c = (rng.uniform(-20, 20), rng.uniform(-2.5, 6), rng.uniform(6, 60))
        rad = rng.uniform(0.15, 0.6)
        kd = (rng.uniform(0.1, 1), rng.uniform(0.1, 1), rng.uniform(0.1, 1))
        refl = rng.uniform(0, 1)
        opaque = rng.uniform() < 0.7
        tr_draw = rng.uniform(0.3, 0.95)
        tr = 0.0 if opaque else tr_draw
        ri = rng.uniform(1.05, 1.5)
        prims.append(sphere(c, rad, kd, reflectivity=refl, transparency=tr, refraction_index=ri))
    lights = _ring_lights(rng, n_lights, 18.0, 30.0)
    return _stack(prims, PRIM_DTYPE), _stack(lights, LIGHT_DTYPE)


def glass_scene(n_spheres: int = 24, n_lights: int = 16, seed: int = 0x61A55):
    """Refraction-heavy scene: ground plane + transparent spheres in front of the camera."""
    rng = SplitMix64(seed)
    prims = [plane((0, -3, 0), (0, 1, 0), (0.5, 0.5, 0.5), reflectivity=0.0, transparency=0.0)]
    for _ in range(n_spheres):
        c = (rng.uniform(-6, 6), rng.uniform(-2, 3.5), rng.uniform(8, 22))
        rad = rng.uniform(0.6, 1.4)
        kd = (rng.uniform(0.2, 1), rng.uniform(0.2, 1), rng.uniform(0.2, 1))
        refl = rng.uniform(0.2, 0.9)
        tr = rng.uniform(0.8, 0.95)
        ri = rng.uniform(1.3, 1.6)
        prims.append(sphere(c, rad, kd, reflectivity=refl, transparency=tr, refraction_index=ri))
    lights = _ring_lights(rng, n_lights, 10.0, 14.0, 0.03, 0.12)
    return _stack(prims, PRIM_DTYPE), _stack(lights, LIGHT_DTYPE)


def orbit_scene(prims: np.ndarray, lights: np.ndarray, angle_rad: float, pivot=(1.75, 0.0, 11.5)):
    """Camera orbit about a vertical axis through `pivot`, applied as the inverse rigid
    rotation of the scene so that the camera stays the reference's (origin, +z).
    Plane normals are rotated and re-normalised the way the Plane constructor would."""
    c, s = math.cos(-angle_rad), math.sin(-angle_rad)

    def rot(v, about):
        x, y, z = v[0] - about[0], v[1] - about[1], v[2] - about[2]
        return (c * x + s * z + about[0], y + about[1], -s * x + c * z + about[2])

    pr, lt = prims.copy(), lights.copy()
    for i in range(len(pr)):
        pr[i]["p"] = rot(pr[i]["p"], pivot)
        if pr[i]["type"] == RT_PRIM_PLANE:
            pr[i]["q"] = make_normal(rot(pr[i]["q"], (0.0, 0.0, 0.0)))
    for i in range(len(lt)):
        lt[i]["pos"] = rot(lt[i]["pos"], pivot)
    return pr, lt
