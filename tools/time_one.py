import sys, os; sys.path.insert(0,'.')
import numpy as np, torch
from raytracer_amd import abi, scenes, device
if os.environ.get("RT_LIB_OVERRIDE"): device.LIB_PATH = os.environ["RT_LIB_OVERRIDE"]
from raytracer_amd.device import DeviceTracer
t = DeviceTracer(0)
def timeit(name, prims, lights, p, n=20):
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
timeit("default 4K D5", pr, lt, abi.make_params(3840,2160,5), 100)
timeit("default 4K D2", pr, lt, abi.make_params(3840,2160,2), 100)
