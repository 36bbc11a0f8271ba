"""rt_render of the headline frame into a caller's ordinary (pageable) numpy array and into a
page-locked one, the array allocated once: median ms per call."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer, HostFrame
t = DeviceTracer(0)
pr, lt = scenes.default_scene(); t.upload_scene(pr, lt)
p = abi.make_params(3840, 2160, 5)
def timed(arr, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); t.render_into(p, arr); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[3:])), float(min(ts[3:]))
a = np.zeros((2160, 3840, 3), dtype=np.uint8)
print("ordinary memory: median %.3f ms, best %.3f" % timed(a))
hf = HostFrame((2160, 3840, 3))
print("page-locked:     median %.3f ms, best %.3f" % timed(hf.array))
assert np.array_equal(a, hf.array)
b = np.zeros((2160, 3840, 3), dtype=np.uint8)
t.render_into(p, b); assert np.array_equal(a, b)
