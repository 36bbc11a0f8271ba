"""The primary-only pass (pixels finished in single precision) against the double-precision walk:
byte differences per scene, and how many pixels the pass finishes.
usage: python tools/front_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer


def with_flags(p, f):
    q = p.copy()
    q["flags"] = int(q["flags"]) | f
    return q


def main():
    t = DeviceTracer(0)
    cases = [("default", scenes.default_scene(), (1280, 720, 5)), ("default4k", scenes.default_scene(), (3840, 2160, 5)),
             ("glass6", scenes.glass_scene(n_spheres=6, n_lights=3), (640, 360, 5)),
             ("plane_only", scenes.glass_scene(n_spheres=0, n_lights=2), (640, 360, 4))]
    for name, (prims, lights), (w, h, d) in cases:
        t.upload_scene(prims, lights)
        p = abi.make_params(w, h, d)
        fast = t.render(p, want_ids=False, want_masks=False)["bgr"].astype(np.int16)
        slow = t.render(with_flags(p, abi.RT_FLAG_NOFRONT), want_ids=False, want_masks=False)["bgr"].astype(np.int16)
        diff = np.abs(fast - slow)
        same = (diff.max(axis=2) == 0).mean()
        print(f"{name}: {w}x{h} D{d}: identical pixels {same:.6f}, largest difference {diff.max()}", flush=True)


if __name__ == "__main__":
    main()
