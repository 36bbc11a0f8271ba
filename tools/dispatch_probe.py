"""How fast the chip gets through waves that do next to nothing: a 4K frame of the default scene
moved behind the camera (every primary ray misses), rendered (a) with the screen bounds - a
wave stores the background and ends, (b) without them but with the single-precision pass - a wave
classifies its primary rays, (c) with neither - a wave normalises its rays in double and searches
the scene.  usage: python tools/dispatch_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracer_amd import abi, scenes  # noqa: E402
from raytracer_amd.device import DeviceTracer  # noqa: E402

t = DeviceTracer(0)
prims, lights = scenes.default_scene()
prims = prims.copy()
pp, qq = prims["p"].copy(), prims["q"].copy()
pp[:, 2] -= 1000.0           # the spheres far behind the camera
pp[0] = (0.0, 0.0, -1000.0)  # the floor: a plane behind the camera, facing it
qq[0] = (0.0, 0.0, 1.0)
prims["p"], prims["q"] = pp, qq
t.upload_scene(prims, lights)
w, h = 3840, 2160
out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
for name, flags in (("screen bounds", 0), ("single-precision pass", abi.RT_FLAG_NOSKY),
                    ("walk", abi.RT_FLAG_NOSKY | abi.RT_FLAG_NOFRONT)):
    p = abi.make_params(w, h, 5)
    p["flags"] = flags
    for _ in range(3):
        t.render_device(p, st.cuda_stream, out.data_ptr())
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            t.render_device(p, st.cuda_stream, out.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    print(f"all-sky 4K frame, {name}: {best * 1000:.1f} us per frame, {w * h / 64 / best / 1e6:.2f} waves per ns", flush=True)
