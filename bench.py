#!/usr/bin/env python3
"""Benchmark of the Whitted trace path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode strips|batch] [--workload NAME]

A step is one frame.  The default workload is the configuration the headline metric is
quoted on: the reference's default scene at 3840x2160, recursion depth 5.

  N = 1       a step is one frame, the whole frame on one GPU.
  N > 1       launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py
              --gpus N ...`, one rank per GPU.
              mode strips (default): a step is a batch of N frames, frame j owned by rank
              j.  Every frame is cut into interleaved 8-row bands, every rank renders its
              strip of each of the N frames (one batched launch, a frame's worth of pixels) and
              ONE RCCL all-to-all moves strip (rank r, frame j) to rank j: the strip gather
              for N frames at once with a different root per frame, so that all xGMI links
              carry the same share.  The exchange of batch b overlaps the render of batch
              b+1.  Work per GPU and step is fixed: weak scaling.
              mode gather: a step is ONE frame, strips gathered to rank 0 (the latency
              path; total work fixed: strong scaling).
              mode batch: a step is one whole frame of a camera orbit per rank, no
              collective (weak scaling).

Rank 0 prints ONE JSON line.  value = rays of all frames of all ranks / wall time of the
timed steps (max over ranks), with scene and buffers resident in HBM before the clock
starts.  Rays = TraceRay invocations that pass the depth stop, counted by the kernel's
counter variant in an untimed launch (and equal to the CPU checker's count).

roofline: the path is bound by the vector ALUs, not by MFMA or HBM, so `bound` is "valu".
`peak` is the gfx950 FP64 vector peak, 78.6 TFLOP/s (256 CUs x 4 SIMDs x 16 lanes x 2 flop x
2.4 GHz; the path computes in f64 - `dtype`); `frac_of_fp32_packed_peak` gives the same
figure against the 157.3 TFLOP/s packed-FP32 vector peak.  achieved = algorithmic
flops of one launch / mean kernel time measured with HIP events around every timed
launch, on the stream the kernel runs on.  Algorithmic flops = event counts x a fixed
cost table (see FLOP_TABLE).  `traffic` is the HBM bytes per launch taken from the PMC
profile committed under profiles/ (null when none matches the workload).

cpu_baseline: the single-thread CPU checker (oracle/, a port of the reference's path)
timed on this host on whole frames of the same workload, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from raytracer_amd import abi, scenes, strips  # noqa: E402

PEAK_FP64_VALU_TFLOPS = 78.6
PEAK_FP32_PACKED_VALU_TFLOPS = 157.3

# Flops per event (mul/add/sub/div/sqrt = 1 each), the reference's arithmetic as written.
FLOP_TABLE = {
    "sphere_tests": 17,      # L, tca, d2, compare (src/objects.h:125-130)
    "plane_tests": 14,       # two dots, a subtraction, a division (src/objects.h:92-96)
    "pixels": 14,            # primary ray: pixel centre + normalisation
    "hits": 37 + 30 + 16 + 45,  # hit setup, reflection ray, refraction k, combine
    "refractions": 26,       # refraction ray when k > 0
    "feelers": 45,           # feeler direction + diffuse term
    "feelers_occluded": 25,  # occluder normal + transmission factor
}

WORKLOADS = {
    # name: (scene builder, width, height, max_depth)
    "default_4k_d5": (scenes.default_scene, 3840, 2160, 5),
    "default_640_d4": (scenes.default_scene, 640, 640, 4),
    "spheres1024_4k_d5": (scenes.random_spheres_scene, 3840, 2160, 5),
    "glass_4k_d8": (scenes.glass_scene, 3840, 2160, 8),
    "default_8k_d5": (scenes.default_scene, 7680, 4320, 5),
}


def algorithmic_flops(counters, pixels: int) -> float:
    f = FLOP_TABLE["pixels"] * pixels
    for k, c in FLOP_TABLE.items():
        if k != "pixels":
            f += c * int(counters[k])
    return float(f)


def cpu_baseline(prims, lights, base, budget_s: float = 12.0):
    """Time the CPU checker on whole frames of the workload (or on a strided subset of rows
    when one frame would not fit the budget)."""
    from tests import oracle_lib

    h = int(base["height"])
    # probe on every 64th row to size the sample
    probe = abi.make_params(int(base["width"]), h, int(base["max_depth"]), band_rows=1, band_first=0,
                            band_step=64, band_count=-(-h // 64))
    t0 = time.perf_counter()
    r = oracle_lib.render(prims, lights, probe, want_ids=False, want_masks=False, want_counters=True)
    dt = time.perf_counter() - t0
    est_frame = dt * 64
    if est_frame <= budget_s:
        # whole frames until the budget is used (the probe only decides between whole frames
        # and sampled rows; it includes the library's first-call costs and is not trusted
        # for more than that)
        p = abi.make_params(int(base["width"]), h, int(base["max_depth"]))
        rays, frames = 0, 0
        t0 = time.perf_counter()
        while frames == 0 or time.perf_counter() - t0 < budget_s:
            r = oracle_lib.render(prims, lights, p, want_ids=False, want_masks=False, want_counters=True)
            rays += int(r["counters"]["rays"])
            frames += 1
        dt = time.perf_counter() - t0
        sample = f"{frames} whole frame(s) of the workload"
        ms_frame = dt / frames * 1e3
    else:
        step = max(2, int(math.ceil(est_frame / budget_s)))
        p = abi.make_params(int(base["width"]), h, int(base["max_depth"]), band_rows=1, band_first=0,
                            band_step=step, band_count=-(-h // step))
        t0 = time.perf_counter()
        r = oracle_lib.render(prims, lights, p, want_ids=False, want_masks=False, want_counters=True)
        dt = time.perf_counter() - t0
        rays = int(r["counters"]["rays"])
        sample = f"every {step}th row of one frame ({abi.strip_rows(p)} of {h} rows)"
        ms_frame = dt * step * 1e3
    return {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": 1,
        "kind": "port",
        "sample": sample,
        "ms_per_frame": ms_frame,
        "cpu": _cpu_model(),
    }


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _traffic_bytes(workload: str):
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=("strips", "gather", "batch"), default="strips")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="default_4k_d5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump", default=None, help="rank 0: write the last frame as BMP to this path")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracer_amd.device import DeviceTracer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and not os.environ.get("RT_BENCH_SHARE_GPU"):
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
    local_rank %= n_dev  # RT_BENCH_SHARE_GPU=1: rehearsal of the N>1 code on fewer GPUs
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RT_BENCH_BACKEND", "nccl")  # "gloo": rehearsal only (host-staged exchange)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    builder, width, height, depth = WORKLOADS[args.workload]
    prims, lights = builder()
    base = abi.make_params(width, height, depth)
    tracer = DeviceTracer(local_rank)
    tracer.upload_scene(prims, lights)
    stream = torch.cuda.current_stream()

    # ---- what this rank renders per step ------------------------------------------------
    if world == 1:
        mode, scaling, frames_per_step = "single", "weak", 1
        my = base.copy()
    elif args.mode == "strips":
        mode, scaling, frames_per_step = "strips", "weak", world
        my = strips.rank_params(base, rank, world)
    elif args.mode == "gather":
        mode, scaling, frames_per_step = "gather", "strong", 1
        my = strips.rank_params(base, rank, world)
    else:
        mode, scaling, frames_per_step = "batch", "weak", world
        my = base.copy()
    my_rows = abi.strip_rows(my)

    # untimed counter launch: rays and event counts of this rank's share of one frame
    d_cnt = torch.zeros(8, dtype=torch.int64, device=device)
    scratch = torch.zeros((max(my_rows, 1), width, 3), dtype=torch.uint8, device=device)
    pc = my.copy()
    pc["flags"] = abi.RT_FLAG_COUNTERS
    tracer.render_device(pc, stream.cuda_stream, scratch.data_ptr(), d_counters=d_cnt.data_ptr())
    torch.cuda.synchronize()
    cnt_host = d_cnt.cpu().numpy().astype(np.uint64)
    counters = {f: int(cnt_host[i]) for i, f in enumerate(abi.COUNTER_FIELDS)}
    my_flops = algorithmic_flops(counters, my_rows * width)  # of ONE launch of this rank
    my_rays = counters["rays"]

    # HIP events around the launches of every timed step, on the streams they run on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    step_no = [0]
    rays_done = [0]  # rays this rank traced in the timed steps
    renders_per_launch = world if mode == "strips" else 1  # strips: one batched launch of N strips

    def timed(i):
        return 0 <= i < args.steps

    if mode in ("single", "gather"):

        def render_strip(params, out):
            i = step_no[0] - args.warmup
            if timed(i):
                ev[i][0].record(stream)
            tracer.render_device(params, stream.cuda_stream, out.data_ptr())
            if timed(i):
                ev[i][1].record(stream)

        gatherer = strips.StripGatherer(base, render_strip, device)

        def step():
            slot = gatherer.submit()
            if timed(step_no[0] - args.warmup):
                rays_done[0] += my_rays
            step_no[0] += 1
            return slot

    elif mode == "strips":
        # one batched launch per step: this rank's strip of each of the N frames
        from raytracer_amd.device import BatchPlan

        plans = {}

        def render_strips(slot, params, outs):
            i = step_no[0] - args.warmup
            if slot not in plans:
                plans[slot] = BatchPlan([tracer] * world, [params] * world, [o.data_ptr() for o in outs])
            if timed(i):
                ev[i][0].record(stream)
            plans[slot].launch(stream.cuda_stream)
            if timed(i):
                ev[i][1].record(stream)

        gatherer = strips.StripExchange(base, render_strips, device)

        def step():
            slot = gatherer.submit()
            if timed(step_no[0] - args.warmup):
                rays_done[0] += my_rays * world  # this rank's strip of each of the N frames
            step_no[0] += 1
            return slot

    else:
        # orbit batch: this rank's frames are different views of the scene; the scene of
        # every view is uploaded before the clock starts.
        gatherer = None
        n_views = 8
        views, view_rays = [], []
        for v in range(n_views):
            t = DeviceTracer(local_rank)
            pr, lt = scenes.orbit_scene(prims, lights, 2 * math.pi * (rank * n_views + v) / (world * n_views))
            t.upload_scene(pr, lt)
            d_cnt.zero_()
            t.render_device(pc, stream.cuda_stream, scratch.data_ptr(), d_counters=d_cnt.data_ptr())
            torch.cuda.synchronize()
            view_rays.append(int(d_cnt[0].item()))
            views.append(t)
        out = [torch.zeros((height, width, 3), dtype=torch.uint8, device=device) for _ in range(2)]

        def step():
            i = step_no[0] - args.warmup
            if timed(i):
                ev[i][0].record(stream)
            views[step_no[0] % n_views].render_device(base, stream.cuda_stream, out[step_no[0] & 1].data_ptr())
            if timed(i):
                ev[i][1].record(stream)
                rays_done[0] += view_rays[step_no[0] % n_views]
            step_no[0] += 1
            return 0

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if gatherer is not None:
        gatherer.drain()
    fence()
    t0 = time.perf_counter()
    last = 0
    for _ in range(args.steps):
        last = step()
    if gatherer is not None:
        gatherer.drain()
    fence()
    elapsed = time.perf_counter() - t0

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")
    my_flops *= renders_per_launch
    my_rows *= renders_per_launch

    # max over ranks of the wall time, sum over ranks of the rays
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_done[0]], dtype=torch.int64, device=device)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = int(r.item())
    else:
        rays_total = rays_done[0]

    if rank == 0:
        if args.dump and gatherer is not None:
            from raytracer_amd.frame import write_bmp

            write_bmp(args.dump, gatherer.frame(last).cpu().numpy())
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_total / elapsed / 1e6
        achieved = my_flops / (kernel_ms * 1e-3) / 1e12
        fb_bytes = my_rows * width * 3
        line = {
            "metric": "Mrays/s (primary+secondary rays traced)",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {width}x{height}, max_depth {depth}, {len(prims)} objects, "
                            f"{len(lights)} lights",
                "mode": mode,
                "backend": (dist.get_backend() if world > 1 else None),
                "frames_per_step": frames_per_step,
                "rays_per_step": rays_total / max(args.steps, 1),
                "band_rows": strips.DEFAULT_BAND_ROWS,
            },
            "roofline": {
                "bound": "valu",
                "achieved": achieved,
                "peak": PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_VALU_TFLOPS,
                "frac_of_fp32_packed_peak": achieved / PEAK_FP32_PACKED_VALU_TFLOPS,
                "traffic": _traffic_bytes(args.workload) if world == 1 else None,
                # template arguments: mask words, counters, sphere tree, deep stack
                "kernel": ("whitted_batch_kernel<%s,%s>" if mode == "strips" else "whitted_kernel<0,false,%s,%s>")
                % (("true" if len(prims) - int((prims["type"] == abi.RT_PRIM_PLANE).sum()) >= 40 else "false"),
                   ("true" if depth > 5 else "false")),
                "kernel_ms": kernel_ms,
                "algorithmic_flops_per_launch": my_flops,
                "framebuffer_write_GBps": fb_bytes / (kernel_ms * 1e-3) / 1e9,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(prims, lights, base)
            line["config"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
