"""ctypes loader of the CPU checker (oracle/_build/liboracle.so) for the tests, the smoke
test and the benchmark's cpu_baseline leg.  Never imported by the raytracer_amd package."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "whitted_oracle.c")
    hdr = os.path.join(ROOT, "include", "rt_hip.h")
    stale = (
        force
        or not os.path.exists(ORACLE_SO)
        or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return ORACLE_SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(ORACLE_SO)
        vp = ctypes.c_void_p
        L.oracle_render.restype = ctypes.c_int
        L.oracle_render.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int32, vp, vp, vp, vp, vp, vp]
        L.oracle_intersect.restype = ctypes.c_int
        L.oracle_intersect.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_int32, vp, vp, vp]
        L.oracle_occluder.restype = ctypes.c_int
        L.oracle_occluder.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp,
                                      ctypes.c_int32, vp, vp]
        L.oracle_nearest.restype = ctypes.c_int
        L.oracle_nearest.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int32, vp, vp]
        L.oracle_strip_rows.restype = ctypes.c_int32
        L.oracle_strip_rows.argtypes = [vp]
        L.oracle_mask_words.restype = ctypes.c_int32
        L.oracle_mask_words.argtypes = [ctypes.c_int32]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def render(prims, lights, params, want_ids=True, want_masks=True, want_counters=False):
    """Run the checker on one partition.  Returns dict(bgr, hit_id, hit_mask, spawn_mask,
    counters) with the shapes rt_render documents."""
    L = lib()
    prims = np.ascontiguousarray(prims, dtype=abi.PRIM_DTYPE)
    lights = np.ascontiguousarray(lights, dtype=abi.LIGHT_DTYPE)
    p = np.array(params, dtype=abi.PARAMS_DTYPE)
    if want_counters:
        p["flags"] = int(p["flags"]) | abi.RT_FLAG_COUNTERS
    rows = L.oracle_strip_rows(_ptr(p))
    if rows < 0:
        raise ValueError("oracle: bad params")
    w = int(p["width"])
    words = abi.mask_words(int(p["max_depth"]))
    bgr = np.zeros((rows, w, 3), dtype=np.uint8)
    hit_id = np.zeros((rows, w), dtype=np.int32) if want_ids else None
    hit_mask = np.zeros((rows, w, words), dtype=np.uint32) if want_masks else None
    spawn_mask = np.zeros((rows, w, words), dtype=np.uint32) if want_masks else None
    counters = np.zeros((), dtype=abi.COUNTERS_DTYPE) if want_counters else None
    rc = L.oracle_render(
        _ptr(prims), len(prims), _ptr(lights), len(lights), _ptr(p),
        _ptr(bgr), _ptr(hit_id), _ptr(hit_mask), _ptr(spawn_mask), _ptr(counters),
    )
    if rc != 0:
        raise ValueError(f"oracle_render failed: {rc}")
    return dict(bgr=bgr, hit_id=hit_id, hit_mask=hit_mask, spawn_mask=spawn_mask, counters=counters)


def intersect(prims, rays, prim):
    L = lib()
    prims = np.ascontiguousarray(prims, dtype=abi.PRIM_DTYPE)
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    n = len(rays)
    t = np.zeros(n, dtype=np.float64)
    hit = np.zeros(n, dtype=np.uint8)
    unit = np.zeros((n, 3), dtype=np.float64)
    rc = L.oracle_intersect(_ptr(prims), len(prims), _ptr(rays), n, prim, _ptr(t), _ptr(hit), _ptr(unit))
    if rc != 0:
        raise ValueError(f"oracle_intersect failed: {rc}")
    return t, hit, unit


def occluder(prims, lights, light, self_index, points):
    """First occluder (index or -1) and occlusion factor of the feelers from `points`."""
    L = lib()
    prims = np.ascontiguousarray(prims, dtype=abi.PRIM_DTYPE)
    lights = np.ascontiguousarray(lights, dtype=abi.LIGHT_DTYPE)
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    n = len(points)
    first = np.zeros(n, dtype=np.int32)
    factor = np.zeros(n, dtype=np.float64)
    rc = L.oracle_occluder(_ptr(prims), len(prims), _ptr(lights), len(lights), light, self_index, _ptr(points), n,
                           _ptr(first), _ptr(factor))
    if rc != 0:
        raise ValueError(f"oracle_occluder failed: {rc}")
    return first, factor


def nearest(prims, rays):
    """Index of the nearest object along each ray (-1 = none) and its t."""
    L = lib()
    prims = np.ascontiguousarray(prims, dtype=abi.PRIM_DTYPE)
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    n = len(rays)
    idx = np.zeros(n, dtype=np.int32)
    t = np.zeros(n, dtype=np.float64)
    rc = L.oracle_nearest(_ptr(prims), len(prims), _ptr(rays), n, _ptr(idx), _ptr(t))
    if rc != 0:
        raise ValueError(f"oracle_nearest failed: {rc}")
    return idx, t
