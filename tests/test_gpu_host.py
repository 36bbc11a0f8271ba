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
    cnt = torch.zeros(8, dtype=torch.int64, device=dev)
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


def test_strip_exchange_and_stream_fan_on_gpu(tracer):
    """The launch pattern of the multi-GPU bench on one GPU: the strips of a frame, cut for
    a world of 4, are rendered as the 4 launches of a batch over side streams, joined into
    the main stream and re-assembled there; three batches reuse both buffer sets."""
    import torch

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    world = 4
    base = abi.make_params(640, 203, 5)
    dev = torch.device("cuda", 0)
    main = torch.cuda.current_stream()
    fan = strips.StreamFan(torch, dev, main, 3)
    mr = strips.max_strip_rows(base, world)
    src = torch.from_numpy(strips.frame_gather_map(base, world)).to(dev)
    bufs = [torch.zeros((world, mr, 640, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    want = tracer.render(base, want_ids=False, want_masks=False)["bgr"]
    for batch in range(3):
        buf = bufs[batch & 1]
        buf.fill_(7)  # on `main`: the launches must not start before this
        for j in range(world):
            p = strips.rank_params(base, j, world)
            st = fan.stream_for(j)
            tracer.render_device(p, st.cuda_stream, buf[j, : abi.strip_rows(p)].data_ptr())
        fan.join()
        frame = buf.reshape(-1, 640, 3).index_select(0, src)  # on `main`, after the join
        assert np.array_equal(frame.cpu().numpy(), want), batch

    # the same through StripExchange with a single rank (no collective)
    def render_strip(j, params, out):
        st = fan.stream_for(j)
        tracer.render_device(params, st.cuda_stream, out.data_ptr())
        fan.join()

    x = strips.StripExchange(base, render_strip, dev)
    slots = [x.submit() for _ in range(3)]
    assert np.array_equal(x.frame(slots[-1]).cpu().numpy(), want)
