import sys, os; sys.path.insert(0,'.')
import numpy as np, torch
from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer
t = DeviceTracer(0)
def timeit(name, prims, lights, p, n=50):
    t.upload_scene(prims, lights)
    rows = abi.strip_rows(p); w = int(p['width'])
    out = torch.zeros((rows, w, 3), dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream()
    for _ in range(3): t.render_device(p, st.cuda_stream, out.data_ptr())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): t.render_device(p, st.cuda_stream, out.data_ptr())
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/n:.4f} ms/frame", flush=True)
pr, lt = scenes.default_scene()
for d in (2,3,5):
    p = abi.make_params(3840,2160,d); timeit(f"fast  D{d}", pr, lt, p)
    p = abi.make_params(3840,2160,d, flags=abi.RT_FLAG_EXACT); timeit(f"exact D{d}", pr, lt, p)
# sky only: one small sphere behind the camera, no lights
far = pr[1:2].copy(); far['p'][:,2] = -50.0
for (w,h) in ((3840,2160),(1920,1080),(7680,4320)):
    p = abi.make_params(w,h,5); timeit(f"fast  all-sky {w}x{h}", far, lt[:0], p)
    p = abi.make_params(w,h,5, flags=abi.RT_FLAG_EXACT); timeit(f"exact all-sky {w}x{h}", far, lt[:0], p)
