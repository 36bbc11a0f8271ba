"""Multi-GPU partition and gather logic on CPU ranks (gloo, world_size 2): the strip
arithmetic, the gather and the de-interleave are the code that runs over RCCL on the
GPUs; only the strip renderer is swapped for the CPU checker."""
import os
import socket

import numpy as np
import pytest

from raytracer_amd import abi, scenes, strips


def test_rank_params_partition_every_band_once():
    for (h, br, world) in ((2160, 8, 8), (2160, 8, 4), (4320, 8, 8), (61, 8, 2), (61, 8, 8), (7, 8, 8), (100, 1, 3)):
        base = abi.make_params(32, h, 5)
        seen = np.zeros(h, dtype=int)
        for r in range(world):
            p = strips.rank_params(base, r, world, br)
            rows = abi.strip_image_rows(p)
            assert len(rows) == abi.strip_rows(p) <= strips.max_strip_rows(base, world, br)
            seen[rows] += 1
        assert (seen == 1).all()
        src = strips.frame_gather_map(base, world, br)
        idx = strips.strip_row_index(base, world, br)
        assert (idx[src] == np.arange(h)).all()


def test_assemble_frame_numpy():
    base = abi.make_params(5, 37, 4)
    world, br = 3, 4
    mr = strips.max_strip_rows(base, world, br)
    gathered = np.zeros((world, mr, 5, 3), dtype=np.uint8)
    for r in range(world):
        rows = abi.strip_image_rows(strips.rank_params(base, r, world, br))
        gathered[r, : len(rows)] = (rows % 251)[:, None, None]
    frame = strips.assemble_frame(gathered, base, world, br)
    assert (frame[:, 0, 0] == np.arange(37) % 251).all()


def test_batch_frames_round_robin():
    got = sorted(f for r in range(8) for f in strips.batch_frames(64, r, 8))
    assert got == list(range(64))
    assert strips.batch_frames(64, 3, 8)[:3] == [3, 11, 19]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, width, height, depth, frames, out_path):
    import torch
    import torch.distributed as dist

    from tests import oracle_lib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prims, lights = scenes.default_scene()
        base = abi.make_params(width, height, depth)
        frame_no = [0]

        def render_strip(params, out):
            # a different scene per frame, so that a buffer mix-up between frames shows
            pr, lt = scenes.orbit_scene(prims, lights, 0.2 * frame_no[0])
            r = oracle_lib.render(pr, lt, params, want_ids=False, want_masks=False)
            out.copy_(torch.from_numpy(r["bgr"]))
            frame_no[0] += 1

        g = strips.StripGatherer(base, render_strip, torch.device("cpu"), band_rows=8)
        assert g.world == world and g.rank == rank
        slots = [g.submit() for _ in range(frames)]  # two in flight, then buffers are reused
        got = g.frame(slots[-1]).numpy().copy() if rank == 0 else None
        g.drain()
        if rank == 0:
            pr, lt = scenes.orbit_scene(prims, lights, 0.2 * (frames - 1))
            want = oracle_lib.render(pr, lt, base, want_ids=False, want_masks=False)["bgr"]
            np.save(out_path, np.array([int(np.array_equal(got, want)), got.shape[0]]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("height", [50, 64])
def test_strip_gather_two_ranks_gloo(tmp_path, height):
    import torch.multiprocessing as mp

    out = str(tmp_path / "result.npy")
    mp.spawn(_worker, args=(2, _free_port(), 72, height, 4, 3, out), nprocs=2, join=True)
    ok, rows = np.load(out)
    assert rows == height and ok == 1


def test_strip_gatherer_single_process():
    import torch

    from tests import oracle_lib

    prims, lights = scenes.default_scene()
    base = abi.make_params(40, 30, 4)

    def render_strip(params, out):
        out.copy_(torch.from_numpy(oracle_lib.render(prims, lights, params, want_ids=False, want_masks=False)["bgr"]))

    g = strips.StripGatherer(base, render_strip, torch.device("cpu"))
    s = g.submit()
    want = oracle_lib.render(prims, lights, base, want_ids=False, want_masks=False)["bgr"]
    assert np.array_equal(g.frame(s).numpy(), want)


def _exchange_worker(rank, world, port, width, height, depth, batches, out_dir):
    import torch
    import torch.distributed as dist

    from tests import oracle_lib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prims, lights = scenes.default_scene()
        base = abi.make_params(width, height, depth)
        batch_no = [-1]

        def view(batch, slot):  # a different scene per frame of every batch
            return scenes.orbit_scene(prims, lights, 0.15 * (batch * world + slot))

        def render_strips(slot, params, outs):
            batch_no[0] += 1
            assert slot == batch_no[0] % 2 and len(outs) == world
            for j, out in enumerate(outs):
                pr, lt = view(batch_no[0], j)
                out.copy_(torch.from_numpy(oracle_lib.render(pr, lt, params, want_ids=False, want_masks=False)["bgr"]))

        x = strips.StripExchange(base, render_strips, torch.device("cpu"), band_rows=8)
        assert x.world == world and x.rank == rank
        slots = [x.submit() for _ in range(batches)]
        got = x.frame(slots[-1]).numpy().copy()
        x.drain()
        pr, lt = view(batches - 1, rank)  # this rank owns frame slot `rank` of the batch
        want = oracle_lib.render(pr, lt, base, want_ids=False, want_masks=False)["bgr"]
        np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([int(np.array_equal(got, want)), got.shape[0]]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("height", [50, 64])
def test_strip_exchange_two_ranks_gloo(tmp_path, height):
    """Batch of `world` frames, strips exchanged with one all-to-all: every rank ends up
    with the whole frame it owns, three batches deep so that both buffer sets are reused."""
    import torch.multiprocessing as mp

    mp.spawn(_exchange_worker, args=(2, _free_port(), 72, height, 4, 3, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok, rows = np.load(tmp_path / f"r{r}.npy")
        assert rows == height and ok == 1


def test_strip_exchange_single_process():
    import torch

    from tests import oracle_lib

    prims, lights = scenes.default_scene()
    base = abi.make_params(40, 30, 4)

    def render_strips(slot, params, outs):
        outs[0].copy_(torch.from_numpy(oracle_lib.render(prims, lights, params, want_ids=False, want_masks=False)["bgr"]))

    x = strips.StripExchange(base, render_strips, torch.device("cpu"))
    s = x.submit()
    want = oracle_lib.render(prims, lights, base, want_ids=False, want_masks=False)["bgr"]
    assert np.array_equal(x.frame(s).numpy(), want)
