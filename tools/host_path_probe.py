"""rt_render (host buffers, PCIe copy included) against rt_render_device on the bench workload."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer
t = DeviceTracer(0)
pr, lt = scenes.default_scene(); t.upload_scene(pr, lt)
p = abi.make_params(3840, 2160, 5)
for _ in range(3): r = t.render(p, want_ids=False, want_masks=False)
n = 20
t0 = time.perf_counter()
for _ in range(n): r = t.render(p, want_ids=False, want_masks=False)
dt = (time.perf_counter() - t0) / n
rays = 20139306
print(f"rt_render 4K D5 bgr only (pageable host buffer): {dt*1e3:.3f} ms/frame, {rays/dt/1e6:.0f} Mrays/s, {24.9e6/dt/1e9:.1f} GB/s of framebuffer")
t0 = time.perf_counter()
for _ in range(5): r = t.render(p)
dt = (time.perf_counter() - t0) / 5
print(f"rt_render 4K D5 with hit_id + masks: {dt*1e3:.3f} ms/frame")
