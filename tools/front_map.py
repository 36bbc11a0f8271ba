"""Which pixels the primary-only pass finishes: renders with an analysis build of the library
that paints them (make the build with -DRT_FRONT_MARK) and prints their share per band of rows.
usage: RT_LIB_OVERRIDE=raytracer_amd/csrc/_ab/librt_mark.so python tools/front_map.py [WORKLOAD]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import abi, device  # noqa: E402

device.LIB_PATH = os.path.abspath(os.environ["RT_LIB_OVERRIDE"])
import bench  # noqa: E402
from raytracer_amd.device import DeviceTracer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "default_4k_d5"
builder, w, h, d = bench.WORKLOADS[name]
prims, lights = builder()
t = DeviceTracer(0)
t.upload_scene(prims, lights)
img = t.render(abi.make_params(w, h, min(d, 5)), want_ids=False, want_masks=False)["bgr"]
mark = (img[..., 0] == 255) & (img[..., 1] == 0) & (img[..., 2] == 255)
print(f"{name}: {mark.mean():.4f} of the pixels finished by the pass")
step = h // 27
for y0 in range(0, h, step):
    band = mark[y0:y0 + step]
    print(f"rows {y0:5d}..{min(y0 + step, h) - 1:5d}: {band.mean():.4f}")
