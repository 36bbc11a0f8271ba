#!/usr/bin/env python3
"""Benchmark of the Whitted trace path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode all|batch|gather|strips]
                  [--workload NAME] [--no-extra] [--no-cpu-baseline] [--no-end-to-end]

A step is one pass of the trace path over one batch of synthetic input.  The default workload
is the configuration the headline metric is quoted on: the reference's default scene at
3840x2160, recursion depth 5.

  N = 1   a step is one frame, the whole frame on one GPU, output resident in HBM.
  N > 1   launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`,
          one rank per GPU.  Three ways to use N GPUs are timed one after the other (K steps
          each, own warm-up) and reported under "modes":
          batch   a step is one whole frame of a camera orbit per rank, no collective (weak
                  scaling; the multi-frame mode).  This is the mode `value` is taken from.
          gather  a step is ONE frame cut into interleaved 8-row bands, every rank renders its
                  strip and the strips are gathered to rank 0 over RCCL (the single-frame mode;
                  total work fixed: strong scaling).  The re-assembly pass on rank 0
                  (index_select of the gathered strips) is part of the timed step.
          strips  a step is a batch of N frames; every rank renders its strip of each of them
                  with one batched launch and ONE RCCL all-to-all moves strip (rank r, frame j)
                  to rank j, which re-assembles its frame (weak scaling).

Rank 0 prints ONE JSON line.  value = rays of all frames of all ranks / wall time of the timed
steps (max over ranks), with scene and buffers resident in HBM before the clock starts.  Rays =
TraceRay invocations that pass the depth stop, counted by the kernel's counter variant in an
untimed launch (and equal to the CPU checker's count) - of the ray tree the timed kernel really
walks: when only the image is asked for, a child under a material coefficient of exactly zero
(reflectivity 0, transparency 0) is not traced, since the reference multiplies what it returns
by zero; the counts are taken with RT_FLAG_PRUNE, which describes that walk.  The reference's
full ray tree of the same frame is reported beside it (`reference_ray_tree`: its ray count and
the rate that count would give), and `gpu_over_cpu` compares frames per second.

Kernel times (`roofline.kernel_ms*`) come from HIP events recorded on the kernel's stream around
every fourth timed launch (`--event-every`): a pair of events costs the stream about 6 us, which
is 5 % of the headline step, so bracketing every launch would slow the timed region it measures.

roofline: the path is bound by the vector ALUs, not by MFMA or HBM, so `bound` is "valu" and
`peak` is the gfx950 FP32 vector peak, 157.3 TFLOP/s.  achieved = algorithmic flops of one
launch / median kernel time measured with HIP events around timed launches, on the stream the
kernel runs on.  Algorithmic flops = event counts x a fixed cost table (FLOP_TABLE): the
reference's arithmetic as written for the events of the walk the kernel makes (a hit whose
children are not walked is not credited with their set-up).  A workload whose kernel searches
the scene through an acceleration structure (sphere tree, anchor grids: from 8 spheres on) never
executes the reference's brute-force tests, so there `achieved` / `frac` are the rate of the
vector lane-operations the kernel really executed (PMC profile of the same launch) against the
unfused rate, 78.6 T/s, and the brute-force count is reported as `equivalent_*`.
`frac_of_fp64_rate` gives the algorithmic figure against 78.6 TFLOP/s, the rate at which the chip
issues the unfused double operations the path computes in.  `traffic` and the
`executed_builder_profile` figures come from the PMC profiles the builder committed under
profiles/ (measured on the builder's box, not in this run; null when none matches the workload).

cpu_baseline: the single-thread CPU checker (oracle/, a C port of the reference's path)
timed on this host on whole frames of the same workload, rank 0 at N=1 only.
end_to_end (N=1): rt_render of the same frame into host memory (ordinary and page-locked), and
the whole program (`rt_main`: process start, scene, render, BMP file) beside the CPU port's.
extra (N=1): the 1024-sphere and the glass workloads, a few steps each, with their own
roofline and sampled CPU baseline.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from raytracer_amd import abi, scenes, strips  # noqa: E402

PEAK_FP32_VALU_TFLOPS = 157.3   # gfx950 vector peak (packed FP32 FMA)
RATE_FP64_VALU_TFLOPS = 78.6    # 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz

# Flops per event (mul/add/sub/div/sqrt = 1 each), the reference's arithmetic as written.
FLOP_TABLE = {
    "sphere_tests": 17,      # L, tca, d2, compare (src/objects.h:125-130)
    "plane_tests": 14,       # two dots, a subtraction, a division (src/objects.h:92-96)
    "pixels": 14,            # primary ray: pixel centre + normalisation
    "hits": 37 + 45,         # hit setup (src/main.cc:55-65) and combine (src/main.cc:100-112)
    "reflections": 9 + 30,   # V / |dot| and the reflection ray, where that child is walked (src/main.cc:70-79)
    "refraction_tests": 16,  # k, where it is evaluated (src/main.cc:86)
    "refraction_rays": 26,   # the refraction ray, where that child is walked (src/main.cc:88-93)
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
EXTRA_WORKLOADS = ("spheres1024_4k_d5", "glass_4k_d8")
# Event counts are taken of the ray tree the timed (image-only) kernel walks: children under a
# material coefficient of zero are left out there (the reference traces them and multiplies by
# zero), and counting them would credit the kernel with rays it never traces.
COUNT_FLAGS = abi.RT_FLAG_COUNTERS | abi.RT_FLAG_PRUNE
LATE_MIN_LIGHTS = 4    # kLateMinLights of the kernel launcher
TREE_MIN_SPHERES = 40  # kTreeMinSpheres
GRID_MIN_SPHERES = 8   # kGridMinSpheres


def algorithmic_flops(counters, pixels: int) -> float:
    ev = {k: int(counters[k]) for k in abi.COUNTER_FIELDS}
    ev["pixels"] = pixels
    # every ray but the primary ones is a reflection or a refraction child that is walked
    ev["refraction_rays"] = ev["rays"] - pixels - ev["reflections"]
    return float(sum(c * ev[k] for k, c in FLOP_TABLE.items()))


def n_spheres(prims) -> int:
    return len(prims) - int((prims["type"] == abi.RT_PRIM_PLANE).sum())


def kernel_name(prims, lights, depth: int, batched: bool) -> str:
    """The instantiation the launcher picks, spelt as the profiler prints it."""
    b = lambda v: "true" if v else "false"
    tree = n_spheres(prims) >= TREE_MIN_SPHERES
    grid = n_spheres(prims) >= GRID_MIN_SPHERES
    deep = depth > 5
    late = (not tree) and (not grid) and len(lights) >= LATE_MIN_LIGHTS
    if batched:
        return f"void rt::whitted_batch_kernel<{b(tree)}, {b(deep)}, {b(late)}, {b(grid)}>(rt::KBatch, int)"
    return f"void rt::whitted_kernel<0, false, {b(tree)}, {b(deep)}, {b(late)}, {b(grid)}>(rt::KArgs, rt::SkyTable)"


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
        "note": "C restatement of the reference's path (the reference itself needs Windows.h and gmtl and cannot "
                "be built here); per ray it is faster than the reference's C++ (no virtual calls, no by-value "
                "material copies), so the ratio is conservative",
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


def _profile_record(workload: str) -> dict:
    """Per-launch PMC figures of the workload's kernel, from profiles/traffic.json."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}) or {}
    except (OSError, ValueError):
        return {}


def roofline(prims, lights, depth, batched, flops, kernel_ms_list, fb_bytes, workload, with_profile=True):
    ms = np.asarray(kernel_ms_list, dtype=np.float64)
    med = float(np.median(ms)) if ms.size else float("nan")  # (a stray slow launch does not move it)
    algorithmic = flops / (med * 1e-3) / 1e12
    prof = _profile_record(workload) if with_profile else {}
    accelerated = n_spheres(prims) >= GRID_MIN_SPHERES  # sphere tree and / or anchor grids
    out = {
        "bound": "valu",
        "achieved": algorithmic,
        "peak": PEAK_FP32_VALU_TFLOPS,
        "unit": "TFLOP/s",
        "frac": algorithmic / PEAK_FP32_VALU_TFLOPS,
        "frac_of_fp64_rate": algorithmic / RATE_FP64_VALU_TFLOPS,
        "traffic": prof.get("hbm_bytes_per_launch"),
        "traffic_from": "builder profile: " + prof["source"] if prof.get("source") else None,
        "kernel": kernel_name(prims, lights, depth, batched),
        "kernel_ms": med,
        "kernel_ms_mean": float(ms.mean()) if ms.size else None,
        "kernel_ms_min": float(ms.min()) if ms.size else None,
        "kernel_ms_max": float(ms.max()) if ms.size else None,
        "algorithmic_flops_per_launch": flops,
        "framebuffer_write_GBps": fb_bytes / (med * 1e-3) / 1e9,
    }
    executed = None
    if prof.get("valu_insts_per_launch"):
        # what the kernel really executed (PMC pass of the same launch on the builder's box)
        lane_ops = prof["valu_insts_per_launch"] * 64.0 * prof.get("lane_utilisation", 1.0)
        executed = {
            "valu_wave_insts_per_launch": prof["valu_insts_per_launch"],
            "lane_utilisation": prof.get("lane_utilisation"),
            "valu_busy": prof.get("valu_busy"),
            "lane_ops_per_s": lane_ops / (med * 1e-3),
            "source": prof.get("source"),
        }
        out["executed_builder_profile"] = executed
    if accelerated:
        # the brute-force tests the event counts stand for are never executed: report them as an
        # equivalent rate and rate the kernel on what it executes
        out["equivalent_TFLOPs"] = algorithmic
        out["equivalent_frac"] = algorithmic / PEAK_FP32_VALU_TFLOPS
        out["equivalent_flops_per_launch"] = flops
        rate = executed["lane_ops_per_s"] / 1e12 if executed else None
        out["achieved"] = rate
        out["peak"] = RATE_FP64_VALU_TFLOPS
        out["unit"] = "T lane-ops/s (executed vector lane-operations, builder profile, against the unfused rate)"
        out["frac"] = rate / RATE_FP64_VALU_TFLOPS if rate is not None else None
        out["frac_of_fp64_rate"] = out["frac"]
        del out["algorithmic_flops_per_launch"]
    return out


class Timed:
    """HIP events around every launch of the timed steps, on the stream the kernel runs on."""

    def __init__(self, torch, stream, steps, warmup, every=1):
        self.stream, self.steps, self.warmup, self.every = stream, steps, warmup, max(1, every)
        self.ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        self.step_no = 0

    def index(self):
        i = self.step_no - self.warmup
        return i if 0 <= i < self.steps else -1

    def launch(self, fn):
        i = self.index()
        timed = i >= 0 and i % self.every == 0
        if timed:
            self.ev[i][0].record(self.stream)
        fn()
        if timed:
            self.ev[i][1].record(self.stream)

    def kernel_ms(self):
        return [a.elapsed_time(b) for i, (a, b) in enumerate(self.ev) if i % self.every == 0]


def count_events(torch, tracer, params, flags, stream, scratch, d_cnt):
    """One untimed launch of the counting kernel: the event counts of the render."""
    pc = params.copy()
    pc["flags"] = flags
    d_cnt.zero_()
    tracer.render_device(pc, stream.cuda_stream, scratch.data_ptr(), d_counters=d_cnt.data_ptr())
    torch.cuda.synchronize()
    c = d_cnt.cpu().numpy().astype(np.uint64)
    return {f: int(c[i]) for i, f in enumerate(abi.COUNTER_FIELDS)}


def run_mode(mode, ctx):
    """Time `steps` steps of one mode.  Returns this rank's figures; collectives (barrier,
    max of the wall time, sum of the rays) are done here so that every rank takes part."""
    torch, dist = ctx["torch"], ctx["dist"]
    world, rank, device, stream = ctx["world"], ctx["rank"], ctx["device"], ctx["stream"]
    tracer, base, args = ctx["tracer"], ctx["base"], ctx["args"]
    prims, lights = ctx["prims"], ctx["lights"]
    width, height = int(base["width"]), int(base["height"])
    timed = Timed(torch, stream, args.steps, args.warmup, args.event_every)
    rays_done = [0]

    if mode in ("single", "gather", "strips"):
        my = base.copy() if mode == "single" else strips.rank_params(base, rank, world)
    else:
        my = base.copy()
    my_rows = abi.strip_rows(my)

    # untimed counter launch: rays and event counts of this rank's share of one frame
    d_cnt = torch.zeros(len(abi.COUNTER_FIELDS), dtype=torch.int64, device=device)
    scratch = torch.zeros((max(my_rows, 1), width, 3), dtype=torch.uint8, device=device)
    pc = my.copy()
    pc["flags"] = COUNT_FLAGS
    tracer.render_device(pc, stream.cuda_stream, scratch.data_ptr(), d_counters=d_cnt.data_ptr())
    torch.cuda.synchronize()
    cnt_host = d_cnt.cpu().numpy().astype(np.uint64)
    counters = {f: int(cnt_host[i]) for i, f in enumerate(abi.COUNTER_FIELDS)}
    my_flops = algorithmic_flops(counters, my_rows * width)  # of ONE render of this rank
    my_rays = counters["rays"]
    reference_rays = None
    if mode == "single":
        # the reference's full ray tree of the same frame, for comparison only
        reference_rays = count_events(torch, tracer, my, abi.RT_FLAG_COUNTERS, stream, scratch, d_cnt)["rays"]
    renders_per_launch = 1
    gatherer = None
    frames_per_step = 1
    scaling = "weak"

    if mode in ("single", "gather"):
        scaling = "weak" if mode == "single" else "strong"

        def render_strip(params, out):
            timed.launch(lambda: tracer.render_device(params, stream.cuda_stream, out.data_ptr()))

        gatherer = strips.StripGatherer(base, render_strip, device)

        prev = [None]

        def step():
            slot = gatherer.submit()
            # the de-interleaved frame is what a step delivers; the previous step's is put
            # together while this step's gather runs
            if mode == "gather" and rank == 0 and prev[0] is not None:
                gatherer.frame(prev[0])
            prev[0] = slot
            if timed.index() >= 0:
                rays_done[0] += my_rays
            timed.step_no += 1
            return slot

    elif mode == "strips":
        from raytracer_amd.device import BatchPlan

        frames_per_step = world
        renders_per_launch = world
        plans = {}

        def render_strips(slot, params, outs):
            if slot not in plans:
                plans[slot] = BatchPlan([tracer] * world, [params] * world, [o.data_ptr() for o in outs])
            timed.launch(lambda: plans[slot].launch(stream.cuda_stream))

        gatherer = strips.StripExchange(base, render_strips, device)

        prev = [None]

        def step():
            slot = gatherer.submit()
            if prev[0] is not None:
                gatherer.frame(prev[0])  # this rank's frame of the previous batch, re-assembled
            prev[0] = slot
            if timed.index() >= 0:
                rays_done[0] += my_rays * world  # this rank's strip of each of the N frames
            timed.step_no += 1
            return slot

    else:  # batch: this rank's frames are different views of the scene, uploaded before the clock starts
        from raytracer_amd.device import DeviceTracer

        frames_per_step = world
        n_views = 8
        views, view_rays, view_flops = [], [], []
        for v in range(n_views):
            t = DeviceTracer(ctx["local_rank"])
            pr, lt = scenes.orbit_scene(prims, lights, 2 * math.pi * (rank * n_views + v) / (world * n_views))
            t.upload_scene(pr, lt)
            d_cnt.zero_()
            t.render_device(pc, stream.cuda_stream, scratch.data_ptr(), d_counters=d_cnt.data_ptr())
            torch.cuda.synchronize()
            c = d_cnt.cpu().numpy().astype(np.uint64)
            view_rays.append(int(c[0]))
            view_flops.append(algorithmic_flops({f: int(c[i]) for i, f in enumerate(abi.COUNTER_FIELDS)}, height * width))
            views.append(t)
        my_flops = float(np.mean(view_flops))
        out = [torch.zeros((height, width, 3), dtype=torch.uint8, device=device) for _ in range(2)]

        def step():
            v = timed.step_no % n_views
            timed.launch(lambda: views[v].render_device(base, stream.cuda_stream, out[timed.step_no & 1].data_ptr()))
            if timed.index() >= 0:
                rays_done[0] += view_rays[v]
            timed.step_no += 1
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

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_done[0]], dtype=torch.int64, device=device)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = int(r.item())
    else:
        rays_total = rays_done[0]

    if rank == 0 and args.dump and gatherer is not None:
        from raytracer_amd.frame import write_bmp

        write_bmp(args.dump, gatherer.frame(last).cpu().numpy())
    return {
        "mode": mode,
        "scaling": scaling,
        "frames_per_step": frames_per_step,
        "elapsed": elapsed,
        "rays_total": rays_total,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "value": rays_total / elapsed / 1e6,
        "kernel_ms": timed.kernel_ms(),
        "flops_per_launch": my_flops * renders_per_launch,
        "fb_bytes": my_rows * renders_per_launch * width * 3,
        "batched": mode == "strips",
        "reference_rays_per_frame": reference_rays,
    }


def end_to_end(tracer, prims, lights, base, cpu_ms_per_frame):
    """rt_render (render + copy to the host) of the workload's frame, and the whole program."""
    from raytracer_amd.device import HostFrame
    from raytracer_amd.frame import write_bmp

    w, h, d = int(base["width"]), int(base["height"]), int(base["max_depth"])
    out = {}

    def time_into(arr, n=12):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            tracer.render_into(base, arr)
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts[2:])), float(min(ts[2:]))

    pageable = np.zeros((h, w, 3), dtype=np.uint8)
    med, lo = time_into(pageable)
    out["rt_render_ms_pageable"] = med
    hf = HostFrame((h, w, 3))
    med, lo = time_into(hf.array)
    out["rt_render_ms_page_locked"] = med
    out["rt_render_ms_page_locked_min"] = lo
    hf.close()
    if cpu_ms_per_frame:
        out["cpu_port_ms_per_frame"] = cpu_ms_per_frame
        out["rt_render_page_locked_over_cpu_port"] = cpu_ms_per_frame / out["rt_render_ms_page_locked"]
        out["rt_render_pageable_over_cpu_port"] = cpu_ms_per_frame / out["rt_render_ms_pageable"]
    # the whole program, as the reference times it (src/main.cc:211,236-238): screen, scene,
    # Tracer, present (here: BMP file).  GPU side: the C++ host program; CPU side: the port.
    exe = os.path.join(ROOT, "raytracer_amd", "host", "rt_main")
    with tempfile.TemporaryDirectory() as tmp:
        if os.path.exists(exe):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "--width", str(w), "--height", str(h), "--depth", str(d), "--frames", "1", "--out",
                                os.path.join(tmp, "gpu.bmp")], capture_output=True, text=True, timeout=300)
            wall = time.perf_counter() - t0
            if r.returncode == 0:
                out["main_equivalent_s_gpu_process_wall"] = wall
                for line in r.stdout.splitlines():
                    if line.startswith("Time:"):
                        out["main_equivalent_s_gpu"] = float(line.split()[1])
                    if line.startswith("Tracer:"):
                        out["tracer_call_s_gpu"] = float(line.split()[1])
            # a sequence of frames: what the first Tracer() call of a process is made of, and what
            # every later call costs (src/main.cc:226-231 called once per frame)
            r = subprocess.run([exe, "--width", str(w), "--height", str(h), "--depth", str(d), "--frames", "3", "--out",
                                os.path.join(tmp, "gpu.bmp")], capture_output=True, text=True, timeout=300)
            if r.returncode == 0:
                for line in r.stdout.splitlines():
                    if line.startswith("TracerFirst:"):
                        out["tracer_first_call_s"] = float(line.split()[1])
                        out["tracer_first_call_parts"] = line.split("(", 1)[1].rstrip(")")
                    if line.startswith("TracerNext:"):
                        out["tracer_next_call_s"] = float(line.split()[1])
        if cpu_ms_per_frame:
            from tests import oracle_lib

            t0 = time.perf_counter()
            r = oracle_lib.render(prims, lights, base, want_ids=False, want_masks=False)
            write_bmp(os.path.join(tmp, "cpu.bmp"), r["bgr"])
            out["main_equivalent_s_cpu_port"] = time.perf_counter() - t0
    return out


def extra_workload(name, torch, device, stream, local_rank, steps=3, warmup=1, cpu_budget_s=6.0):
    """A few steps of another workload on this GPU, with its own roofline and CPU sample."""
    from raytracer_amd.device import DeviceTracer

    builder, width, height, depth = WORKLOADS[name]
    prims, lights = builder()
    base = abi.make_params(width, height, depth)
    tracer = DeviceTracer(local_rank)
    tracer.upload_scene(prims, lights)
    out = torch.zeros((height, width, 3), dtype=torch.uint8, device=device)
    d_cnt = torch.zeros(len(abi.COUNTER_FIELDS), dtype=torch.int64, device=device)
    pc = base.copy()
    pc["flags"] = COUNT_FLAGS
    tracer.render_device(pc, stream.cuda_stream, out.data_ptr(), d_counters=d_cnt.data_ptr())
    torch.cuda.synchronize()
    c = d_cnt.cpu().numpy().astype(np.uint64)
    counters = {f: int(c[i]) for i, f in enumerate(abi.COUNTER_FIELDS)}
    flops = algorithmic_flops(counters, width * height)
    for _ in range(warmup):
        tracer.render_device(base, stream.cuda_stream, out.data_ptr())
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        tracer.render_device(base, stream.cuda_stream, out.data_ptr())
        b.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms = [a.elapsed_time(b) for a, b in ev]
    rec = {
        "workload": f"{name}: {width}x{height}, max_depth {depth}, {len(prims)} objects, {len(lights)} lights",
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "value": counters["rays"] * steps / elapsed / 1e6,
        "unit": "Mrays/s",
        "rays_per_step": counters["rays"],
        "roofline": roofline(prims, lights, depth, False, flops, kms, width * height * 3, name),
        "cpu_baseline": cpu_baseline(prims, lights, base, budget_s=cpu_budget_s),
    }
    ref = count_events(torch, tracer, base, abi.RT_FLAG_COUNTERS, stream, out, d_cnt)["rays"]
    rec["reference_ray_tree"] = {"rays_per_frame": ref, "Mrays_per_s_equivalent": ref * steps / elapsed / 1e6}
    rec["gpu_over_cpu"] = rec["cpu_baseline"]["ms_per_frame"] / rec["ms_per_step"]  # frames per second
    tracer.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=("all", "strips", "gather", "batch"), default="all")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="default_4k_d5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="N=1: skip the 1024-sphere and glass workloads")
    ap.add_argument("--no-end-to-end", action="store_true", help="N=1: skip the rt_render / rt_main timings")
    ap.add_argument("--event-every", type=int, default=4,
                    help="HIP events around every n-th timed launch (1 = every launch): an event pair costs the "
                         "stream about 6 us, 5 %% of the headline step")
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
    ctx = dict(torch=torch, dist=dist, world=world, rank=rank, local_rank=local_rank, device=device, stream=stream,
               tracer=tracer, base=base, args=args, prims=prims, lights=lights)

    if world == 1:
        modes = ["single"]
    elif args.mode == "all":
        modes = ["batch", "gather", "strips"]  # `value` comes from the first one
    else:
        modes = [args.mode]
    # what every rank runs on, as the process group sees it (rank 0 prints it)
    me = f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}"
    if world > 1:
        names = [None] * world
        dist.all_gather_object(names, me)
        ctx["rank_devices"] = names
    else:
        ctx["rank_devices"] = [me]
    results = [run_mode(m, ctx) for m in modes]

    if rank == 0:
        head = results[0]
        line = {
            "metric": "Mrays/s (primary+secondary rays traced)",
            "value": head["value"],
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": head["scaling"],
            "vs_baseline": None,
            "dtype": "f64 decisions (f32 where the answer is certain at that precision) / f32 colour",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {width}x{height}, max_depth {depth}, {len(prims)} objects, "
                            f"{len(lights)} lights",
                "mode": head["mode"],
                "backend": (dist.get_backend() if world > 1 else None),
                "frames_per_step": head["frames_per_step"],
                "rays_per_step": head["rays_total"] / max(args.steps, 1),
                "band_rows": strips.DEFAULT_BAND_ROWS,
                "rays_counted": "the ray tree the image-only kernel walks: children under a material coefficient "
                                "of zero (reflectivity 0 / transparency 0) contribute zero and are not traced",
                "single_precision_paths": "shadow feelers of launches that use the anchor grids, and pixels whose "
                                          "primary ray ends the ray tree, are answered in single precision where every "
                                          "test is certain there (otherwise in double); their rays, tests and "
                                          "flops are counted like the others'",
            },
            "roofline": roofline(prims, lights, depth, head["batched"], head["flops_per_launch"], head["kernel_ms"],
                                 head["fb_bytes"], args.workload, with_profile=(world == 1)),
        }
        if world > 1:
            line["config"]["rccl_ranks"] = dist.get_world_size()
            line["config"]["rank_devices"] = ctx["rank_devices"]
            line["modes"] = {
                r["mode"]: {
                    "value": r["value"], "unit": "Mrays/s", "ms_per_step": r["ms_per_step"], "scaling": r["scaling"],
                    "frames_per_step": r["frames_per_step"], "kernel_ms_rank0": float(np.mean(r["kernel_ms"])),
                }
                for r in results
            }
        if world == 1:
            ref = head["reference_rays_per_frame"]
            line["config"]["reference_ray_tree"] = {
                "rays_per_frame": ref,
                "Mrays_per_s_equivalent": ref / (head["ms_per_step"] * 1e-3) / 1e6,
            }
            cpu_ms = None
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(prims, lights, base)
                # frames per second (the CPU traces the reference's full ray tree)
                line["config"]["gpu_over_cpu"] = line["cpu_baseline"]["ms_per_frame"] / line["ms_per_step"]
                cpu_ms = line["cpu_baseline"]["ms_per_frame"]
            if not args.no_end_to_end:
                line["end_to_end"] = end_to_end(tracer, prims, lights, base, cpu_ms)
            if not args.no_extra and args.workload == "default_4k_d5":
                line["extra"] = {n: extra_workload(n, torch, device, stream, local_rank) for n in EXTRA_WORKLOADS}
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
