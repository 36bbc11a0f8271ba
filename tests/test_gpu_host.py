"""GPU tests of the paths around the kernel: the C++ host program end to end, the
device-pointer entry point used under torch streams, and the strip gatherer on cuda."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import abi, frame, scenes, strips
from tests import oracle_lib
from tests.parity import assert_parity

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_program_matches_checker(tmp_path):
    """rt_main = the reference's main() flow (scene built by the C++ classes, Tracer(),
    Screen, file dump): its BMP equals the checker's frame of the same configuration."""
    exe = os.path.join(ROOT, "raytracer_amd", "host", "rt_main")
    assert os.path.exists(exe), "build raytracer_amd/host first (__graft_entry__.build())"
    for ext in ("bmp", "ppm"):
        out = str(tmp_path / f"f.{ext}")
        r = subprocess.run([exe, "--width", "160", "--height", "90", "--depth", "4", "--out", out],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert "Time:" in r.stdout
    prims, lights = scenes.default_scene()
    ora = oracle_lib.render(prims, lights, abi.make_params(160, 90, 4), want_ids=False, want_masks=False)
    img = frame.read_bmp(str(tmp_path / "f.bmp"))
    assert_parity(dict(bgr=img), ora, "rt_main bmp")
    data = open(str(tmp_path / "f.ppm"), "rb").read()
    head = b"P6\n160 90\n255\n"
    rgb = np.frombuffer(data[len(head):], dtype=np.uint8).reshape(90, 160, 3)
    assert np.array_equal(rgb[::-1, :, ::-1], img)


@pytest.mark.parametrize("scene,depth,orbit", [("glass", 8, 0.0), ("spheres1024", 5, 0.0), ("default", 5, 40.0)])
def test_cpp_host_program_other_scenes(tmp_path, scene, depth, orbit):
    """rt_main on the scenes of the larger configurations and on an orbit view (all built by
    the C++ host's own builders) against the checker on the Python builders' scenes."""
    import math

    exe = os.path.join(ROOT, "raytracer_amd", "host", "rt_main")
    out = str(tmp_path / "f.bmp")
    cmd = [exe, "--scene", scene, "--width", "96", "--height", "64", "--depth", str(depth), "--out", out]
    if orbit:
        cmd += ["--orbit", str(orbit)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    build = {"glass": scenes.glass_scene, "spheres1024": scenes.random_spheres_scene, "default": scenes.default_scene}
    prims, lights = build[scene]()
    if orbit:
        prims, lights = scenes.orbit_scene(prims, lights, orbit * 3.141592653589793 / 180.0)
    ora = oracle_lib.render(prims, lights, abi.make_params(96, 64, depth), want_ids=False, want_masks=False)
    assert_parity(dict(bgr=frame.read_bmp(out)), ora, f"rt_main {scene}")


@pytest.mark.parametrize("scene,size,depth,band_rows", [("default", (640, 363), 5, 8), ("glass", (160, 100), 8, 5)])
def test_cpp_strip_gather_over_rccl(tmp_path, scene, size, depth, band_rows):
    """rt_gather = the multi-GPU single-frame path from C++ (one process, one RCCL communicator
    per GPU of the box: strips rendered with rt_render_device into the buffers ncclGather sends,
    bands put back in frame order on the host): its frame equals rt_main's of the same
    configuration, byte for byte."""
    exe = os.path.join(ROOT, "raytracer_amd", "host", "rt_gather")
    ref = os.path.join(ROOT, "raytracer_amd", "host", "rt_main")
    assert os.path.exists(exe), "build raytracer_amd/host first (__graft_entry__.build())"
    got, want = str(tmp_path / "g.bmp"), str(tmp_path / "m.bmp")
    common = ["--scene", scene, "--width", str(size[0]), "--height", str(size[1]), "--depth", str(depth)]
    r = subprocess.run([exe] + common + ["--band-rows", str(band_rows), "--frames", "2", "--out", got],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Gather:" in r.stdout
    r = subprocess.run([ref] + common + ["--out", want], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(frame.read_bmp(got), frame.read_bmp(want))


def test_render_device_on_torch_stream(tracer):
    import torch

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    p = abi.make_params(200, 120, 5, flags=abi.RT_FLAG_COUNTERS)
    words = abi.mask_words(5)
    dev = torch.device("cuda", 0)
    bgr = torch.zeros((120, 200, 3), dtype=torch.uint8, device=dev)
    hid = torch.zeros((120, 200), dtype=torch.int32, device=dev)
    hm = torch.zeros((120, 200, words), dtype=torch.int32, device=dev)
    sm = torch.zeros((120, 200, words), dtype=torch.int32, device=dev)
    cnt = torch.zeros(len(abi.COUNTER_FIELDS), dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        tracer.render_device(p, side.cuda_stream, bgr.data_ptr(), hid.data_ptr(), hm.data_ptr(), sm.data_ptr(),
                             cnt.data_ptr())
    side.synchronize()
    host = tracer.render(p, want_counters=True)
    assert np.array_equal(bgr.cpu().numpy(), host["bgr"])
    assert np.array_equal(hid.cpu().numpy(), host["hit_id"])
    assert np.array_equal(hm.cpu().numpy().view(np.uint32), host["hit_mask"])
    assert np.array_equal(sm.cpu().numpy().view(np.uint32), host["spawn_mask"])
    got = cnt.cpu().numpy()
    assert [int(v) for v in got] == [int(host["counters"][f]) for f in abi.COUNTER_FIELDS]


def test_strip_gatherer_on_gpu_single_rank(tracer):
    import torch

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    base = abi.make_params(320, 200, 5)
    stream = torch.cuda.current_stream()

    def render_strip(params, out):
        tracer.render_device(params, stream.cuda_stream, out.data_ptr())

    g = strips.StripGatherer(base, render_strip, torch.device("cuda", 0))
    slots = [g.submit() for _ in range(3)]
    got = g.frame(slots[-1]).cpu().numpy()
    want = tracer.render(base, want_ids=False, want_masks=False)["bgr"]
    assert np.array_equal(got, want)


def test_batched_launch_on_gpu(tracer):
    """rt_render_device_batch, the per-step launch of the multi-GPU path, on one GPU: the
    strips of a frame cut for a world of 4 — with a second scene in two of the slots and a
    smaller frame in one — rendered by ONE launch equal the same strips rendered one by
    one; then the frame re-assembled from its four strips equals the whole frame, over
    three rounds that reuse both buffer sets of a StripExchange."""
    import torch

    from raytracer_amd.device import BatchPlan, DeviceTracer

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    other = DeviceTracer(0)
    other.upload_scene(*scenes.glass_scene(n_spheres=5, n_lights=2))
    dev = torch.device("cuda", 0)
    main = torch.cuda.current_stream()
    world = 4
    base = abi.make_params(640, 203, 5)
    small = abi.make_params(96, 50, 7)
    tr = [tracer, other, tracer, other]
    ps = [strips.rank_params(base, 0, world), strips.rank_params(base, 1, world), strips.rank_params(small, 1, 2),
          strips.rank_params(base, 3, world)]
    outs = [torch.zeros((abi.strip_rows(p), int(p["width"]), 3), dtype=torch.uint8, device=dev) for p in ps]
    BatchPlan(tr, ps, [o.data_ptr() for o in outs]).launch(main.cuda_stream)
    for t, p, o in zip(tr, ps, outs):
        want = t.render(p, want_ids=False, want_masks=False)["bgr"]
        assert np.array_equal(o.cpu().numpy(), want)
    # more renders than one launch holds
    many = [strips.rank_params(base, j, 11) for j in range(11)]
    outs = [torch.zeros((abi.strip_rows(p), 640, 3), dtype=torch.uint8, device=dev) for p in many]
    BatchPlan([tracer] * 11, many, [o.data_ptr() for o in outs]).launch(main.cuda_stream)
    for p, o in zip(many, outs):
        assert np.array_equal(o.cpu().numpy(), tracer.render(p, want_ids=False, want_masks=False)["bgr"])

    # a frame as the four strips of one batched launch, through StripExchange's buffers
    want = tracer.render(base, want_ids=False, want_masks=False)["bgr"]
    mr = strips.max_strip_rows(base, world)
    src = torch.from_numpy(strips.frame_gather_map(base, world)).to(dev)
    bufs = [torch.zeros((world, mr, 640, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    ps = [strips.rank_params(base, j, world) for j in range(world)]
    plans = [BatchPlan([tracer] * world, ps, [b[j].data_ptr() for j in range(world)]) for b in bufs]
    for batch in range(3):
        bufs[batch & 1].fill_(7)
        plans[batch & 1].launch(main.cuda_stream)
        frame = bufs[batch & 1].reshape(-1, 640, 3).index_select(0, src)
        assert np.array_equal(frame.cpu().numpy(), want), batch

    # StripExchange with a single rank (no collective), plan per buffer set
    x_plans = {}

    def render_strips(slot, params, outs):
        if slot not in x_plans:
            x_plans[slot] = BatchPlan([tracer], [params], [outs[0].data_ptr()])
        x_plans[slot].launch(main.cuda_stream)

    x = strips.StripExchange(base, render_strips, dev)
    slots = [x.submit() for _ in range(3)]
    assert np.array_equal(x.frame(slots[-1]).cpu().numpy(), want)
    other.close()


def test_deep_renders_on_two_streams(tracer):
    """Depth-8 renders keep part of their per-lane stacks in a global workspace; two of them
    enqueued on different streams of one context may overlap and must not share it."""
    import torch

    prims, lights = scenes.glass_scene()
    tracer.upload_scene(prims, lights)
    p = abi.make_params(320, 200, 8)
    want = tracer.render(p, want_ids=False, want_masks=False)["bgr"]
    dev = torch.device("cuda", 0)
    outs = [torch.zeros((200, 320, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for i, o in enumerate(outs):
        tracer.render_device(p, streams[i & 1].cuda_stream, o.data_ptr())
    for s in streams:
        s.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), want)


def test_render_device_under_graph_capture(tracer):
    """rt_render_device / rt_render_device_batch enqueue one kernel and allocate nothing at
    depths up to 5, so they can be captured in a HIP graph: four frames recorded once,
    replayed twice."""
    import torch

    from raytracer_amd.device import BatchPlan

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    dev = torch.device("cuda", 0)
    ps = [abi.make_params(256, 160, d) for d in (2, 3, 4, 5)]
    outs = [torch.zeros((160, 256, 3), dtype=torch.uint8, device=dev) for _ in ps]
    plan = BatchPlan([tracer] * 2, ps[2:], [o.data_ptr() for o in outs[2:]])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        tracer.render_device(ps[0], st, outs[0].data_ptr())
        tracer.render_device(ps[1], st, outs[1].data_ptr())
        plan.launch(st)
    for _ in range(2):
        for o in outs:
            o.zero_()
        g.replay()
        torch.cuda.synchronize()
        for p, o in zip(ps, outs):
            assert np.array_equal(o.cpu().numpy(), tracer.render(p, want_ids=False, want_masks=False)["bgr"])


def test_sphere_tree_scene_under_graph_capture(tracer):
    """A scene that is rendered through the sphere tree (40 spheres or more) takes its tiles
    from a queue, whose 256 bytes come from a pool the context allocated at rt_create: the
    render and the batched render are captured on a stream the context has never seen,
    without a warm-up launch, and replayed."""
    import torch

    from raytracer_amd.device import BatchPlan, DeviceTracer

    prims, lights = scenes.random_spheres_scene(64, 3, seed=0x64)
    dev = torch.device("cuda", 0)
    with DeviceTracer(0) as fresh:            # a context that has not launched anything yet
        fresh.upload_scene(prims, lights)
        ps = [abi.make_params(256, 160, d) for d in (3, 5, 4, 5)]
        outs = [torch.zeros((160, 256, 3), dtype=torch.uint8, device=dev) for _ in ps]
        plan = BatchPlan([fresh] * 2, ps[2:], [o.data_ptr() for o in outs[2:]])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st = torch.cuda.current_stream().cuda_stream
            fresh.render_device(ps[0], st, outs[0].data_ptr())
            fresh.render_device(ps[1], st, outs[1].data_ptr())
            plan.launch(st)
        for _ in range(2):
            for o in outs:
                o.zero_()
            g.replay()
            torch.cuda.synchronize()
            for p, o in zip(ps, outs):
                assert np.array_equal(o.cpu().numpy(), fresh.render(p, want_ids=False, want_masks=False)["bgr"])
                assert int(o.max().item()) > 0


def test_strip_exchange_over_rccl_single_rank(tracer):
    """The multi-GPU step's call sequence on one GPU: an RCCL process group of one rank,
    batched launch into the send buffer, all_to_all_single (async) on device tensors,
    wait, re-assembly from the receive buffer; buffers reused over four batches."""
    import socket

    import torch
    import torch.distributed as dist

    from raytracer_amd.device import BatchPlan

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        prims, lights = scenes.default_scene()
        tracer.upload_scene(prims, lights)
        base = abi.make_params(640, 203, 5)
        stream = torch.cuda.current_stream()
        plans = {}

        def render_strips(slot, params, outs):
            if slot not in plans:
                plans[slot] = BatchPlan([tracer], [params], [outs[0].data_ptr()])
            plans[slot].launch(stream.cuda_stream)

        x = strips.StripExchange(base, render_strips, dev, exchange_alone=True)
        assert x.collective and not x.staged
        slots = [x.submit() for _ in range(4)]
        x.drain()
        want = tracer.render(base, want_ids=False, want_masks=False)["bgr"]
        for sl in set(slots):
            assert np.array_equal(x.frame(sl).cpu().numpy(), want)
        # the gather path of the single-frame mode, same group
        def render_strip(params, out):
            tracer.render_device(params, stream.cuda_stream, out.data_ptr())

        g = strips.StripGatherer(base, render_strip, dev)
        assert np.array_equal(g.frame(g.submit()).cpu().numpy(), want)
    finally:
        dist.destroy_process_group()


def test_bench_step_functions_under_rccl_single_rank(tracer):
    """bench.py's own per-mode step functions (batch, gather, strips: what `bench.py --gpus N`
    times on every rank) run inside an RCCL process group of one rank: every mode completes,
    counts the rays of the frames it rendered and reports a kernel time."""
    import argparse
    import socket

    import torch
    import torch.distributed as dist

    import bench

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        builder, width, height, depth = bench.WORKLOADS["default_640_d4"]
        prims, lights = builder()
        tracer.upload_scene(prims, lights)
        base = abi.make_params(width, height, depth)
        args = argparse.Namespace(steps=3, warmup=1, event_every=1, dump=None)
        ctx = dict(torch=torch, dist=dist, world=dist.get_world_size(), rank=dist.get_rank(), local_rank=0, device=dev,
                   stream=torch.cuda.current_stream(), tracer=tracer, base=base, args=args, prims=prims, lights=lights)
        for mode in ("batch", "gather", "strips"):
            r = bench.run_mode(mode, ctx)
            assert r["mode"] == mode and r["elapsed"] > 0 and len(r["kernel_ms"]) == 3
            assert r["rays_total"] >= 3 * width * height          # at least the primary rays of three frames
            assert r["flops_per_launch"] > 14 * width * height
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                                        # the group itself works
        assert float(t.sum().item()) == 4.0
    finally:
        dist.destroy_process_group()


def test_rt_render_in_parts_and_into_page_locked_memory(tracer):
    """rt_render of the image alone renders the partition in parts and copies each part while
    the next is rendered: whole frames, strips of bands and a clipped band, into ordinary
    and into page-locked host memory, equal to the one-launch render with the id/mask outputs."""
    from raytracer_amd.device import HostFrame

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    cases = [abi.make_params(1920, 1083, 5), strips.rank_params(abi.make_params(1600, 1203, 4), 1, 3),
             abi.make_params(1024, 700, 5, band_rows=300, band_first=1, band_step=1, band_count=1),
             abi.make_params(1024, 700, 5, band_rows=400, band_first=0, band_step=1, band_count=1)]
    for p in cases:
        ids = tracer.render(p)["bgr"]                       # one launch, not split
        walked = p.copy()
        walked["flags"] = abi.RT_FLAG_NOFRONT              # every pixel walked in double precision:
        assert np.array_equal(tracer.render(walked, want_ids=False, want_masks=False)["bgr"], ids)  # the same bytes
        got = tracer.render(p, want_ids=False, want_masks=False)["bgr"]
        assert_parity({"bgr": got}, {"bgr": ids}, "image only, in parts")
        # ... and the same bytes as the image-only render done with one launch
        import torch

        one = torch.zeros(ids.shape, dtype=torch.uint8, device="cuda")
        tracer.render_device(p, torch.cuda.current_stream().cuda_stream, one.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(got, one.cpu().numpy())
        hf = HostFrame(ids.shape)
        hf.array[:] = 3
        tracer.render_into(p, hf.array)
        assert np.array_equal(hf.array, got)
        hf.close()
    # a depth-8 frame in parts: every part's launch re-arms the tile queue of the workspace
    tracer.upload_scene(*scenes.glass_scene())
    # (an image-only launch of this scene lets the lanes of a wave share rays, which sums a
    # pixel's colour in another order than the launch that also writes ids: colour tolerance)
    p = abi.make_params(1280, 720, 8)
    whole = tracer.render(p)["bgr"]
    first = tracer.render(p, want_ids=False, want_masks=False)["bgr"]
    assert_parity({"bgr": first}, {"bgr": whole}, "glass depth 8 in parts")
    tracer.trim()
    again = tracer.render(p, want_ids=False, want_masks=False)["bgr"]
    assert np.array_equal(again, first)  # a new workspace, the same bytes


def test_rt_render_into_ordinary_memory_with_counters(tracer):
    """rt_render of the image alone into ordinary (not page-locked) memory stages the copy in the
    library (DMA into its own page-locked buffer, copy threads): frames of sizes that do not
    divide into its chunks, whole and as a strip, with the event counters asked for as well,
    equal the render into page-locked memory and the full render's counters."""
    from raytracer_amd.device import HostFrame

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    for p in (abi.make_params(1999, 1001, 4), strips.rank_params(abi.make_params(2560, 1441, 5), 1, 2),
              abi.make_params(3840, 2160, 5)):
        full = tracer.render(p, want_counters=True)
        img = tracer.render(p, want_ids=False, want_masks=False, want_counters=True)  # a fresh numpy array
        shape = img["bgr"].shape
        assert shape[0] * shape[1] * 3 >= (2 << 20)  # large enough for the staged path
        for k in full["counters"].dtype.names:
            assert int(img["counters"][k]) == int(full["counters"][k]), k
        assert np.array_equal(img["bgr"], full["bgr"])  # (the counting kernels walk every pixel)
        hf = HostFrame(shape)
        tracer.render_into(p, hf.array)
        again = np.full(shape, 7, dtype=np.uint8)
        tracer.render_into(p, again)
        assert np.array_equal(again, hf.array)
        hf.close()


@pytest.mark.parametrize("n_ctx,band_rows,size", [(1, 8, (640, 203)), (3, 8, (1920, 1083)), (8, 8, (3840, 2160)),
                                                  (4, 16, (1000, 37)), (5, 8, (64, 20))])
def test_render_node_equals_whole_frame(tracer, n_ctx, band_rows, size):
    """rt_render_node: the frame dealt out in interleaved bands over n contexts (all on the one
    GPU here, each with its own streams and staging), every part copied straight into its rows
    of the caller's frame, equals the whole-frame render."""
    from raytracer_amd.device import DeviceTracer, HostFrame, render_node

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    others = [DeviceTracer(0) for _ in range(n_ctx - 1)]
    for t in others:
        t.upload_scene(prims, lights)
    p = abi.make_params(size[0], size[1], 5)
    want = tracer.render(p, want_ids=False, want_masks=False)["bgr"]
    out = np.full(want.shape, 9, dtype=np.uint8)
    render_node([tracer] + others, p, out, band_rows)
    assert np.array_equal(out, want)
    hf = HostFrame(want.shape)
    hf.array[:] = 1
    render_node([tracer] + others, p, hf.array, band_rows)
    assert np.array_equal(hf.array, want)
    hf.close()
    for t in others:
        t.close()


def test_upload_rejects_what_the_filters_are_not_derived_for(tracer):
    from raytracer_amd.device import RtError

    prims, lights = scenes.default_scene()
    bad = prims.copy()
    bad[0]["q"] = (0.0, 2.0, 0.0)          # a plane normal that is not of unit length
    with pytest.raises(RtError):
        tracer.upload_scene(bad, lights)
    bad = prims.copy()
    bad[0]["q"] = (1000.0, -1000.0, 0.0)
    with pytest.raises(RtError):
        tracer.upload_scene(bad, lights)
    bad = prims.copy()
    bad[1]["reserved"] = 1
    with pytest.raises(RtError):
        tracer.upload_scene(bad, lights)
    # the zero normal (what gmtl leaves of a zero vector) is a plane nothing ever hits
    zero = prims.copy()
    zero[0]["q"] = (0.0, 0.0, 0.0)
    p = abi.make_params(64, 48, 4)
    tracer.upload_scene(zero, lights)
    dev = tracer.render(p, want_counters=True)
    ora = oracle_lib.render(zero, lights, p, want_counters=True)
    assert_parity(dev, ora, "zero normal")
