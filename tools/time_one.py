"""Kernel time of whole-frame renders, HIP events around n back-to-back launches.
usage: [RT_LIB_OVERRIDE=path/to/lib.so] time_one.py [--depth D] [--flags F] [--no-lights] [--rows R] [WORKLOAD ...]
(default: the headline workload; --depth / --flags / --no-lights vary the render: a ladder of
such renders shows where a frame's time goes; --rows R renders only the lowest R image rows)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from raytracer_amd import abi, device  # noqa: E402

if os.environ.get("RT_LIB_OVERRIDE"):
    device.LIB_PATH = os.path.abspath(os.environ["RT_LIB_OVERRIDE"])
import bench  # noqa: E402
from raytracer_amd.device import DeviceTracer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--no-lights", action="store_true")
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("workloads", nargs="*")
args = ap.parse_args()

t = DeviceTracer(0)
for name in (args.workloads or ["default_4k_d5"]):
    builder, w, h, d = bench.WORKLOADS[name]
    prims, lights = builder()
    if args.no_lights:
        lights = lights[:0]
    p = abi.make_params(w, h, args.depth or d)
    p["flags"] = args.flags
    if args.rows:
        p["band_rows"], p["band_count"] = args.rows, 1
    n = 100 if name.startswith("default") else 4
    t.upload_scene(prims, lights)
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    for _ in range(3 if n > 4 else 1):
        t.render_device(p, st.cuda_stream, out.data_ptr())
    torch.cuda.synchronize()
    best = 1e9
    tot = 0.0
    reps = 5 if n > 4 else 2
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            t.render_device(p, st.cuda_stream, out.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        best = min(best, ms)
        tot += ms
    tag = f"{name} depth {args.depth or d} flags {args.flags}{' no lights' if args.no_lights else ''}{f' rows {args.rows}' if args.rows else ''}"
    print(f"{os.path.basename(device.LIB_PATH)} {tag}: mean {tot / reps:.4f} best {best:.4f} ms/frame", flush=True)
