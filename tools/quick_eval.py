"""GPU quick evaluation: parity of the render path vs the CPU checker on a few scenes
and kernel timing of the bench workloads."""
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer
from tests import oracle_lib
t = DeviceTracer(0)
def cmp(name, prims, lights, p, check=True):
    t.upload_scene(prims, lights)
    dev = t.render(p)
    px = dev['bgr'].shape[0]*dev['bgr'].shape[1]
    msg = f"{name}: px {px}"
    if check:
        ora = oracle_lib.render(prims, lights, p)
        bad = [int((dev[k] != ora[k]).sum()) for k in ('hit_id','hit_mask','spawn_mask')]
        diff = np.abs(dev['bgr'].astype(int) - ora['bgr'].astype(int))
        msg += f" bad id/hm/sm {bad} colour maxdiff {diff.max()} px!=0 {(diff.max(-1)>0).mean():.4%}"
    print(msg, flush=True)
pr, lt = scenes.default_scene()
cmp("default 640 D4", pr, lt, abi.make_params(640,640,4))
cmp("default 640 D8", pr, lt, abi.make_params(640,640,8))
cmp("default 4K D5", pr, lt, abi.make_params(3840,2160,5))
gp, gl = scenes.glass_scene()
cmp("glass 256x144 D8", gp, gl, abi.make_params(256,144,8))
sp, sl = scenes.random_spheres_scene()
cmp("spheres 192x108 D5", sp, sl, abi.make_params(192,108,5))
# timing
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
timeit("default 4K D5", pr, lt, abi.make_params(3840,2160,5), 50)
timeit("default 4K D2", pr, lt, abi.make_params(3840,2160,2), 50)
timeit("glass 4K D8", gp, gl, abi.make_params(3840,2160,8), 3)
timeit("spheres 4K D5", sp, sl, abi.make_params(3840,2160,5), 2)
