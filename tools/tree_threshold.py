"""Linear object loop against the sphere tree over scene sizes (one GPU, 1920x1080, depth 5)."""
import sys; sys.path.insert(0, '.')
import torch
from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer

t = DeviceTracer(0)
def timeit(p, n=5):
    rows = abi.strip_rows(p); w = int(p['width'])
    out = torch.zeros((rows, w, 3), dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream()
    t.render_device(p, st.cuda_stream, out.data_ptr()); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): t.render_device(p, st.cuda_stream, out.data_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, (pr, lt), d in [("glass24", scenes.glass_scene(), 8)] + [(f"spheres{n}", scenes.random_spheres_scene(n, 8, seed=0x5EED1024), 5) for n in (8, 16, 32, 64, 128, 256, 1024)]:
    t.upload_scene(pr, lt)
    lin = timeit(abi.make_params(1920, 1080, d, flags=abi.RT_FLAG_LINEAR))
    tree = timeit(abi.make_params(1920, 1080, d, flags=abi.RT_FLAG_TREE))
    print(f"{name}: linear {lin:.3f} ms  tree {tree:.3f} ms  ratio {lin/tree:.2f}", flush=True)
pr, lt = scenes.random_spheres_scene()
t.upload_scene(pr, lt)
print(f"spheres1024 4K D5: linear {timeit(abi.make_params(3840, 2160, 5, flags=abi.RT_FLAG_LINEAR), 2):.1f} ms  tree {timeit(abi.make_params(3840, 2160, 5), 3):.1f} ms ratio", flush=True)
