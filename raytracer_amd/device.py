"""ctypes binding of librt_hip.so (the C ABI in include/rt_hip.h).

There is no CPU fallback: importing works anywhere (so that host logic can be tested
without a GPU), but creating a DeviceTracer without the built library or without a
gfx950 device raises.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_hip.so")

# every symbol include/rt_hip.h declares
EXPORTS = (
    "rt_mask_words",
    "rt_strip_rows",
    "rt_version",
    "rt_create",
    "rt_destroy",
    "rt_last_error",
    "rt_upload_scene",
    "rt_render",
    "rt_render_device",
    "rt_render_device_static",
    "rt_render_device_batch",
    "rt_render_node",
    "rt_trim",
    "rt_host_alloc",
    "rt_host_free",
    "rt_debug_intersect",
    "rt_debug_occluder",
    "rt_debug_nearest",
)

_lib = None


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


def load_library():
    """dlopen librt_hip.so and declare the prototypes.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback path"
        )
    # PyTorch's ROCm wheels carry their own copy of the HIP runtime under the system copy's
    # SONAME; whichever is loaded first serves the whole process, and torch does not find its
    # GPUs on top of the system copy.  So when torch is installed it goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    L.rt_mask_words.restype, L.rt_mask_words.argtypes = i32, [i32]
    L.rt_strip_rows.restype, L.rt_strip_rows.argtypes = i32, [vp]
    L.rt_version.restype, L.rt_version.argtypes = ctypes.c_char_p, []
    L.rt_create.restype, L.rt_create.argtypes = ctypes.c_int, [i32, ctypes.POINTER(vp)]
    L.rt_destroy.restype, L.rt_destroy.argtypes = None, [vp]
    L.rt_last_error.restype, L.rt_last_error.argtypes = ctypes.c_char_p, [vp]
    L.rt_upload_scene.restype, L.rt_upload_scene.argtypes = ctypes.c_int, [vp, vp, i32, vp, i32]
    L.rt_render.restype, L.rt_render.argtypes = ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp]
    L.rt_render_device.restype, L.rt_render_device.argtypes = ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]
    L.rt_render_device_static.restype = ctypes.c_int
    L.rt_render_device_static.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.rt_render_device_batch.restype, L.rt_render_device_batch.argtypes = ctypes.c_int, [vp, vp, i32, vp, vp]
    L.rt_render_node.restype, L.rt_render_node.argtypes = ctypes.c_int, [vp, i32, vp, i32, vp]
    L.rt_trim.restype, L.rt_trim.argtypes = ctypes.c_int, [vp]
    L.rt_host_alloc.restype, L.rt_host_alloc.argtypes = ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(vp)]
    L.rt_host_free.restype, L.rt_host_free.argtypes = None, [vp]
    L.rt_debug_nearest.restype = ctypes.c_int
    L.rt_debug_nearest.argtypes = [vp, vp, i32, ctypes.c_uint32, vp, vp]
    L.rt_debug_occluder.restype = ctypes.c_int
    L.rt_debug_occluder.argtypes = [vp, vp, i32, i32, i32, ctypes.c_uint32, vp, vp]
    L.rt_debug_intersect.restype, L.rt_debug_intersect.argtypes = ctypes.c_int, [vp, vp, i32, i32, vp, vp, vp]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class DeviceTracer:
    """One rt_ctx: a scene resident on one MI355X, rendered strip by strip."""

    def __init__(self, device_id: int = 0):
        self._L = load_library()
        h = ctypes.c_void_p()
        rc = self._L.rt_create(device_id, ctypes.byref(h))
        if rc != abi.RT_OK:
            raise RtError(rc, self._L.rt_last_error(None).decode())
        self._h = h
        self.device_id = device_id

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != abi.RT_OK:
            raise RtError(rc, self._L.rt_last_error(self._h).decode())

    def upload_scene(self, prims: np.ndarray, lights: np.ndarray):
        self._prims = np.ascontiguousarray(prims, dtype=abi.PRIM_DTYPE)
        self._lights = np.ascontiguousarray(lights, dtype=abi.LIGHT_DTYPE)
        self._check(
            self._L.rt_upload_scene(self._h, _ptr(self._prims), len(self._prims), _ptr(self._lights), len(self._lights))
        )

    def render(self, params, want_ids=True, want_masks=True, want_counters=False):
        """rt_render into fresh host arrays.  Same dict as the checker returns."""
        p = np.array(params, dtype=abi.PARAMS_DTYPE)
        if want_counters:
            p["flags"] = int(p["flags"]) | abi.RT_FLAG_COUNTERS
        rows = self._L.rt_strip_rows(_ptr(p))
        if rows < 0:
            raise RtError(rows, "bad rt_params")
        w = int(p["width"])
        words = abi.mask_words(int(p["max_depth"]))
        bgr = np.zeros((rows, w, 3), dtype=np.uint8)
        hit_id = np.zeros((rows, w), dtype=np.int32) if want_ids else None
        hit_mask = np.zeros((rows, w, words), dtype=np.uint32) if want_masks else None
        spawn_mask = np.zeros((rows, w, words), dtype=np.uint32) if want_masks else None
        counters = np.zeros((), dtype=abi.COUNTERS_DTYPE) if want_counters else None
        self._check(
            self._L.rt_render(self._h, _ptr(p), _ptr(bgr), _ptr(hit_id), _ptr(hit_mask), _ptr(spawn_mask), _ptr(counters))
        )
        return dict(bgr=bgr, hit_id=hit_id, hit_mask=hit_mask, spawn_mask=spawn_mask, counters=counters)

    def render_device(self, params, stream: int, d_bgr: int, d_hit_id: int = 0, d_hit_mask: int = 0,
                      d_spawn_mask: int = 0, d_counters: int = 0):
        """rt_render_device: enqueue on `stream` (a hipStream_t as int), outputs are device
        addresses (e.g. torch tensor .data_ptr()).  Returns immediately."""
        p = np.array(params, dtype=abi.PARAMS_DTYPE)
        as_vp = lambda v: ctypes.c_void_p(v) if v else None
        self._check(
            self._L.rt_render_device(
                self._h, _ptr(p), as_vp(stream), as_vp(d_bgr), as_vp(d_hit_id), as_vp(d_hit_mask),
                as_vp(d_spawn_mask), as_vp(d_counters),
            )
        )

    def render_device_static(self, params, stream: int, d_bgr: int, d_hit_id: int = 0, d_hit_mask: int = 0,
                             d_spawn_mask: int = 0, d_counters: int = 0):
        """rt_render_device_static: as render_device, but a deep-stack render uses the
        workspace built into the library, so the call never allocates."""
        p = np.array(params, dtype=abi.PARAMS_DTYPE)
        as_vp = lambda v: ctypes.c_void_p(v) if v else None
        self._check(
            self._L.rt_render_device_static(
                self._h, _ptr(p), as_vp(stream), as_vp(d_bgr), as_vp(d_hit_id), as_vp(d_hit_mask),
                as_vp(d_spawn_mask), as_vp(d_counters),
            )
        )

    def render_into(self, params, out: np.ndarray):
        """rt_render of the image alone into a caller's uint8 array (e.g. a HostFrame's)."""
        p = np.array(params, dtype=abi.PARAMS_DTYPE)
        rows = self._L.rt_strip_rows(_ptr(p))
        if rows < 0 or out.dtype != np.uint8 or out.size != rows * int(p["width"]) * 3 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("render_into: `out` must be a contiguous uint8 array of rows*width*3 bytes")
        self._check(self._L.rt_render(self._h, _ptr(p), _ptr(out), None, None, None, None))
        return out

    def trim(self):
        """rt_trim: wait for the device and free the deep-stack workspaces."""
        self._check(self._L.rt_trim(self._h))

    def debug_intersect(self, rays: np.ndarray, prim: int):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = len(rays)
        t = np.zeros(n, dtype=np.float64)
        hit = np.zeros(n, dtype=np.uint8)
        unit = np.zeros((n, 3), dtype=np.float64)
        self._check(self._L.rt_debug_intersect(self._h, _ptr(rays), n, prim, _ptr(t), _ptr(hit), _ptr(unit)))
        return t, hit, unit

    def debug_nearest(self, rays: np.ndarray, flags: int = 0):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = len(rays)
        idx = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float64)
        self._check(self._L.rt_debug_nearest(self._h, _ptr(rays), n, flags, _ptr(idx), _ptr(t)))
        return idx, t

    def debug_occluder(self, points: np.ndarray, light: int, self_index: int = -1, flags: int = 0):
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        n = len(points)
        first = np.zeros(n, dtype=np.int32)
        factor = np.zeros(n, dtype=np.float64)
        self._check(self._L.rt_debug_occluder(self._h, _ptr(points), n, light, self_index, flags, _ptr(first), _ptr(factor)))
        return first, factor


class BatchPlan:
    """Prebuilt arguments of rt_render_device_batch: n renders in one launch, render i with
    tracers[i]'s scene, params[i], output device address d_bgr[i].  Building the plan once
    and calling launch(stream) per step keeps the per-step host cost to one foreign call."""

    def __init__(self, tracers, params, d_bgr):
        n = len(tracers)
        if not (n == len(params) == len(d_bgr)) or n == 0:
            raise ValueError("BatchPlan: tracers, params and d_bgr must have the same non-zero length")
        self._tracers = list(tracers)  # keep the contexts alive
        self._L = tracers[0]._L
        self._n = n
        self._ctxs = (ctypes.c_void_p * n)(*[t._h for t in tracers])
        self._params = np.ascontiguousarray(np.stack([np.array(p, dtype=abi.PARAMS_DTYPE) for p in params]))
        self._out = (ctypes.c_void_p * n)(*[ctypes.c_void_p(int(a)) for a in d_bgr])

    def launch(self, stream: int):
        rc = self._L.rt_render_device_batch(
            self._ctxs, _ptr(self._params), self._n, ctypes.c_void_p(stream) if stream else None, self._out
        )
        if rc != abi.RT_OK:
            raise RtError(rc, self._L.rt_last_error(self._tracers[0]._h).decode())


class HostFrame:
    """A page-locked host buffer (rt_host_alloc) viewed as a numpy uint8 array of `shape`."""

    def __init__(self, shape):
        self._L = load_library()
        n = int(np.prod(shape))
        p = ctypes.c_void_p()
        rc = self._L.rt_host_alloc(max(n, 1), ctypes.byref(p))
        if rc != abi.RT_OK:
            raise RtError(rc, self._L.rt_last_error(None).decode())
        self._p = p
        self.array = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(max(n, 1),))[:n].reshape(shape)

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self._L.rt_host_free(self._p)
            self._p = None

    __del__ = close


def render_node(tracers, params, out: np.ndarray, band_rows: int = 8):
    """rt_render_node: one frame dealt out in bands of `band_rows` rows over `tracers` (each
    holding the scene), written into `out` ([height, width, 3] uint8, host)."""
    L = tracers[0]._L
    p = np.array(params, dtype=abi.PARAMS_DTYPE)
    h, w = int(p["height"]), int(p["width"])
    if out.dtype != np.uint8 or out.size != h * w * 3 or not out.flags["C_CONTIGUOUS"]:
        raise ValueError("render_node: `out` must be a contiguous uint8 array of height*width*3 bytes")
    ctxs = (ctypes.c_void_p * len(tracers))(*[t._h for t in tracers])
    rc = L.rt_render_node(ctxs, len(tracers), _ptr(p), band_rows, _ptr(out))
    if rc != abi.RT_OK:
        raise RtError(rc, L.rt_last_error(tracers[0]._h).decode())
    return out
