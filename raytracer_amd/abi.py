"""Binary layout of the structs declared in include/rt_hip.h, as numpy dtypes.

The same arrays are handed to the device library and (in tests) to the CPU checker, so
both see identical bytes.  Field order and sizes must match the header exactly; the
sizes are asserted at import.
"""
from __future__ import annotations

import numpy as np

RT_OK = 0
RT_ERR_ARG = -1
RT_ERR_HIP = -2
RT_ERR_STATE = -3
RT_ERR_NOMEM = -4

RT_MAX_DEPTH = 8
RT_MAX_PRIMS = 65536
RT_MAX_LIGHTS = 256
RT_PRIM_PLANE = 0
RT_PRIM_SPHERE = 1
RT_FLAG_COUNTERS = 1
RT_FLAG_LINEAR = 2
RT_FLAG_TREE = 4
RT_FLAG_PRUNE = 8
RT_FLAG_NOGRID = 16
RT_FLAG_GRID = 32
RT_FLAG_NOSKY = 64
RT_FLAG_NOFRONT = 128
RT_DEBUG_RECORDS = 256

PRIM_DTYPE = np.dtype(
    [
        ("type", "<i4"),
        ("reserved", "<i4"),
        ("p", "<f8", (3,)),
        ("q", "<f8", (3,)),
        ("kd", "<f8", (3,)),
        ("reflectivity", "<f8"),
        ("transparency", "<f8"),
        ("refraction_index", "<f8"),
        ("ambient_intensity", "<f8"),
    ]
)
LIGHT_DTYPE = np.dtype([("pos", "<f8", (3,)), ("color", "<f8", (3,)), ("intensity", "<f8")])
PARAMS_DTYPE = np.dtype(
    [
        ("width", "<i4"),
        ("height", "<i4"),
        ("max_depth", "<i4"),
        ("band_rows", "<i4"),
        ("band_first", "<i4"),
        ("band_step", "<i4"),
        ("band_count", "<i4"),
        ("flags", "<u4"),
        ("fov_deg", "<f8"),
        ("bgr", "<f8", (3,)),
    ]
)
COUNTER_FIELDS = (
    "rays",
    "primary_hits",
    "hits",
    "sphere_tests",
    "plane_tests",
    "feelers",
    "feelers_occluded",
    "refractions",
    "reflections",
    "refraction_tests",
)
COUNTERS_DTYPE = np.dtype([(f, "<u8") for f in COUNTER_FIELDS])

assert PRIM_DTYPE.itemsize == 112
assert LIGHT_DTYPE.itemsize == 56
assert PARAMS_DTYPE.itemsize == 64
assert COUNTERS_DTYPE.itemsize == 80


def mask_words(max_depth: int) -> int:
    """Words of one pixel's node mask: max(1, 2^(max_depth-1)/32) (rt_mask_words)."""
    bits = 1 << (max_depth - 1)
    return 1 if bits < 32 else bits // 32


def make_params(
    width: int,
    height: int,
    max_depth: int,
    bgr=(0.58, 0.76, 0.92),
    fov_deg: float = 60.0,
    band_rows: int | None = None,
    band_first: int = 0,
    band_step: int = 1,
    band_count: int | None = None,
    flags: int = 0,
) -> np.ndarray:
    """One rt_params record.  Default partition = the whole frame as a single band."""
    p = np.zeros((), dtype=PARAMS_DTYPE)
    p["width"], p["height"], p["max_depth"] = width, height, max_depth
    p["band_rows"] = height if band_rows is None else band_rows
    p["band_first"], p["band_step"] = band_first, band_step
    if band_count is None:
        band_count = 1 if band_rows is None else -(-height // band_rows)
    p["band_count"] = band_count
    p["flags"] = flags
    p["fov_deg"] = fov_deg
    p["bgr"] = bgr
    return p


def strip_rows(p: np.ndarray) -> int:
    """Rows held by the partition in p after clipping to the frame (rt_strip_rows)."""
    h, br = int(p["height"]), int(p["band_rows"])
    rows = 0
    for b in range(int(p["band_count"])):
        y0 = (int(p["band_first"]) + b * int(p["band_step"])) * br
        if y0 >= h:
            break
        rows += min(y0 + br, h) - y0
    return rows


def strip_image_rows(p: np.ndarray) -> np.ndarray:
    """Image row index of every strip row of the partition in p, in strip order."""
    h, br = int(p["height"]), int(p["band_rows"])
    out = []
    for b in range(int(p["band_count"])):
        y0 = (int(p["band_first"]) + b * int(p["band_step"])) * br
        if y0 >= h:
            break
        out.extend(range(y0, min(y0 + br, h)))
    return np.asarray(out, dtype=np.int64)
