"""One GPU: cost of rendering a frame's worth of pixels as the N strips of one batched launch
(the per-rank work of a multi-GPU step) against one whole-frame launch."""
import sys, time; sys.path.insert(0, '.')
import torch
from raytracer_amd import abi, scenes, strips
from raytracer_amd.device import BatchPlan, DeviceTracer

t = DeviceTracer(0)
pr, lt = scenes.default_scene()
t.upload_scene(pr, lt)
base = abi.make_params(3840, 2160, 5)
dev = torch.device("cuda", 0)
main = torch.cuda.current_stream()

def run(world, steps=300):
    mr = strips.max_strip_rows(base, world)
    buf = torch.zeros((world, mr, 3840, 3), dtype=torch.uint8, device=dev)
    ps = [strips.rank_params(base, j, world) for j in range(world)]
    plan = BatchPlan([t] * world, ps, [buf[j].data_ptr() for j in range(world)])
    for _ in range(20): plan.launch(main.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): plan.launch(main.cuda_stream)
    t_cpu = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"world {world}: {t_all/steps*1e3:.4f} ms/step (cpu enqueue {t_cpu/steps*1e3:.4f})", flush=True)

for w in (1, 2, 4, 8, 16):
    run(w)
