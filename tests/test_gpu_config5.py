"""GPU tests of the multi-GPU configuration's building blocks on one GPU: the 8K frame (whole,
and cut into the strips of a world of 8 through the batched launch), the 64-view camera orbit
dealt out over 8 ranks, and every instantiation of the batched kernel (sphere tree, deep
stack, many lights) against per-strip renders and the CPU checker."""
import math

import numpy as np
import pytest

from raytracer_amd import abi, scenes, strips
from tests import oracle_lib
from tests.parity import assert_parity

pytestmark = pytest.mark.gpu


def _batched(tracers, params, torch):
    """One rt_render_device_batch launch: render i with tracers[i] and params[i]."""
    from raytracer_amd.device import BatchPlan

    dev = torch.device("cuda", 0)
    outs = [torch.full((max(abi.strip_rows(p), 1), int(p["width"]), 3), 5, dtype=torch.uint8, device=dev) for p in params]
    BatchPlan(tracers, params, [o.data_ptr() for o in outs]).launch(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return [o[: abi.strip_rows(p)].cpu().numpy() for o, p in zip(outs, params)]


def test_8k_frame_whole_and_as_strips_of_world_8(tracer):
    """Default scene, 7680x4320, depth 5: the whole frame against the checker pixel by pixel;
    then the frame cut for a world of 8 (interleaved 8-row bands), the eight strips rendered by
    one batched launch and put back together with the gather map, equal to the whole frame."""
    import torch

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    base = abi.make_params(7680, 4320, 5)
    dev = tracer.render(base, want_counters=True)
    ora = oracle_lib.render(prims, lights, base, want_counters=True)
    assert_parity(dev, ora, "default 8K D5")
    world = 8
    ps = [strips.rank_params(base, r, world) for r in range(world)]
    parts = _batched([tracer] * world, ps, torch)
    mr = strips.max_strip_rows(base, world)
    padded = np.zeros((world, mr, 7680, 3), dtype=np.uint8)
    for r, part in enumerate(parts):
        padded[r, : part.shape[0]] = part
    frame = strips.assemble_frame(padded, base, world)
    # the strips are image-only renders: byte for byte the whole frame rendered that way (a pixel
    # does not depend on how the frame is cut), which meets the checker's frame to the tolerance
    whole = tracer.render(base, want_ids=False, want_masks=False)["bgr"]
    assert np.array_equal(frame, whole)
    assert_parity({"bgr": whole}, {"bgr": ora["bgr"]}, "default 8K D5, image only")


def test_orbit_batch_of_64_views_over_8_ranks(tracer):
    """The 64-frame camera orbit at 4K, frames dealt out with strips.batch_frames (no
    collective): every rank's frame list is disjoint and together they cover the orbit; each
    rank's first view is compared with the checker on the whole frame, every view of the orbit
    on a strided subset of rows; views go through the batched launch eight at a time."""
    import torch

    from raytracer_amd.device import DeviceTracer

    n_views, world = 64, 8
    lists = [strips.batch_frames(n_views, r, world) for r in range(world)]
    assert sorted(sum(lists, [])) == list(range(n_views))
    prims, lights = scenes.default_scene()
    rows = abi.make_params(3840, 2160, 5, band_rows=1, band_first=3, band_step=40, band_count=54)
    whole = abi.make_params(3840, 2160, 5)
    for r in range(world):
        views = []
        for f in lists[r]:
            t = DeviceTracer(0)
            pr, lt = scenes.orbit_scene(prims, lights, 2 * math.pi * f / n_views)
            t.upload_scene(pr, lt)
            views.append((t, pr, lt))
        got = _batched([v[0] for v in views], [rows] * len(views), torch)
        for (t, pr, lt), g in zip(views, got):
            ora = oracle_lib.render(pr, lt, rows, want_ids=False, want_masks=False)
            assert_parity(dict(bgr=g), ora, f"orbit rank {r}")
        t, pr, lt = views[0]
        dev = t.render(whole, want_counters=True)
        ora = oracle_lib.render(pr, lt, whole, want_counters=True)
        assert_parity(dev, ora, f"orbit view {lists[r][0]} whole frame")
        for v in views:
            v[0].close()


@pytest.mark.parametrize("case", ["tree", "tree_deep", "late", "deep_late", "deep", "plain"])
def test_batched_kernel_instantiations(tracer, case):
    """rt_render_device_batch with a scene of 48 spheres (sphere tree) or 6 spheres
    (container-order loop), 5 lights (late shadow predicate) or 2, at depth 8 (deep stack) or
    5: the strips of a world of 3 from one launch equal the same strips rendered one by one,
    and the re-assembled frame passes against the checker."""
    import torch

    n_spheres = 48 if case.startswith("tree") else 6
    n_lights = 5 if "late" in case else 2
    depth = 8 if "deep" in case else 5
    prims, lights = scenes.glass_scene(n_spheres=n_spheres, n_lights=n_lights, seed=0xBA7C4 + n_spheres + n_lights)
    tracer.upload_scene(prims, lights)
    base = abi.make_params(200, 117, depth)
    world = 3
    ps = [strips.rank_params(base, r, world) for r in range(world)]
    parts = _batched([tracer] * world, ps, torch)
    for p, part in zip(ps, parts):
        assert np.array_equal(part, tracer.render(p, want_ids=False, want_masks=False)["bgr"])
    mr = strips.max_strip_rows(base, world)
    padded = np.zeros((world, mr, 200, 3), dtype=np.uint8)
    for r, part in enumerate(parts):
        padded[r, : part.shape[0]] = part
    frame = strips.assemble_frame(padded, base, world)
    ora = oracle_lib.render(prims, lights, base, want_ids=False, want_masks=False)
    assert_parity(dict(bgr=frame), ora, f"batched {case}")
