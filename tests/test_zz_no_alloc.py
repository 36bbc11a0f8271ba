"""The entry point that allocates nothing (rt_render_device_static) against rt_render_device:
a depth-8 render keeps part of its ray stacks in a workspace, which the static variant takes
from an array built into the library instead of allocating one per stream.  Two calls in a
row on one stream, the second with a smaller input, every output identical."""
import numpy as np
import pytest

from raytracer_amd import abi, scenes

pytestmark = pytest.mark.gpu


def _render(tracer, static, p, stream, torch):
    dev = torch.device("cuda", 0)
    h, w, words = abi.strip_rows(p), int(p["width"]), abi.mask_words(int(p["max_depth"]))
    bgr = torch.full((h, w, 3), 7, dtype=torch.uint8, device=dev)
    hid = torch.full((h, w), -9, dtype=torch.int32, device=dev)
    hm = torch.zeros((h, w, words), dtype=torch.int32, device=dev)
    sm = torch.zeros((h, w, words), dtype=torch.int32, device=dev)
    cnt = torch.zeros(len(abi.COUNTER_FIELDS), dtype=torch.int64, device=dev)
    q = p.copy()
    q["flags"] = int(q["flags"]) | abi.RT_FLAG_COUNTERS
    fn = tracer.render_device_static if static else tracer.render_device
    fn(q, stream.cuda_stream, bgr.data_ptr(), hid.data_ptr(), hm.data_ptr(), sm.data_ptr(), cnt.data_ptr())
    image = torch.full((h, w, 3), 9, dtype=torch.uint8, device=dev)
    fn(p, stream.cuda_stream, image.data_ptr())  # the image-only kernel
    return bgr, hid, hm, sm, cnt, image


def test_static_workspace_render_equals_allocating_render(tracer):
    import torch

    prims, lights = scenes.glass_scene()
    tracer.upload_scene(prims, lights)
    stream = torch.cuda.Stream()
    big = abi.make_params(320, 200, 8)
    small = abi.make_params(96, 64, 7, band_rows=8, band_first=1, band_step=2, band_count=4)
    with torch.cuda.stream(stream):
        got = [_render(tracer, True, p, stream, torch) for p in (big, small)]   # two calls in a row
        stream.synchronize()
        want = [_render(tracer, False, p, stream, torch) for p in (big, small)]
        stream.synchronize()
    for g, w_, p in zip(got, want, (big, small)):
        for a, b, name in zip(g, w_, ("bgr", "hit_id", "hit_mask", "spawn_mask", "counters", "image")):
            assert np.array_equal(a.cpu().numpy(), b.cpu().numpy()), (name, int(p["width"]))
        assert np.array_equal(g[0].cpu().numpy(), g[5].cpu().numpy())
        assert int(g[4][0].item()) > abi.strip_rows(p) * int(p["width"])  # rays: the frame was really traced
    # and the shallow depths, which use no workspace at all
    p5 = abi.make_params(160, 90, 5)
    a = _render(tracer, True, p5, stream, torch)
    b = _render(tracer, False, p5, stream, torch)
    stream.synchronize()
    for x, y in zip(a, b):
        assert np.array_equal(x.cpu().numpy(), y.cpu().numpy())


def test_static_entry_allocates_nothing(tracer):
    """What the entry point is for: on a stream the context has never seen, renders that need a
    workspace (depth 8; a sphere-tree scene at depth 5, which needs the tile queue only, and at
    depth 8) leave the device's free memory exactly as it was, and give the frames of the
    allocating entry point."""
    import torch

    dev = torch.device("cuda", 0)
    cases = [
        (scenes.glass_scene(), abi.make_params(256, 144, 8)),
        (scenes.random_spheres_scene(48, 3, seed=0x48), abi.make_params(256, 144, 5)),
        (scenes.random_spheres_scene(48, 3, seed=0x48), abi.make_params(256, 144, 8)),
    ]
    for (prims, lights), p in cases:
        tracer.upload_scene(prims, lights)
        h, w = abi.strip_rows(p), int(p["width"])
        out_s = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        out_a = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        fresh = torch.cuda.Stream()       # the context has no workspace for this stream, and never gets one here
        # (the first launch on a new stream lets the HIP runtime set up the stream's queue, its
        # private-memory arena included: that is not the library's doing and is kept out of the count)
        tracer.render_device_static(p, fresh.cuda_stream, out_s.data_ptr())
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        out_s.zero_()
        tracer.render_device_static(p, fresh.cuda_stream, out_s.data_ptr())
        tracer.render_device_static(p, fresh.cuda_stream, out_s.data_ptr())
        fresh.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "rt_render_device_static changed the free device memory"
        tracer.render_device(p, fresh.cuda_stream, out_a.data_ptr())
        fresh.synchronize()
        assert np.array_equal(out_s.cpu().numpy(), out_a.cpu().numpy())
        assert int(out_s.max().item()) > 0
    tracer.trim()
