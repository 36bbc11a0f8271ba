"""Splitting one frame across the GPUs of a node and putting it back together.

Pixels are independent (no state is shared between two TraceRay calls, the scene is
read-only), so a frame is cut into bands of `band_rows` image rows and band b goes to rank
b mod world: interleaving keeps cheap sky rows and expensive object rows on every rank.
Every rank holds the whole scene (kilobytes).  The only exchange is one gather of the
8-bit strips to rank 0 (RCCL over xGMI: every sender uses its own direct link to the root,
so the gather is link-parallel).  Multi-frame batches need no collective at all: frame f
goes to rank f mod world.

Nothing here touches the GPU by itself; the renderer is passed in, which is also how the
logic is tested on CPU ranks over gloo.
"""
from __future__ import annotations

import numpy as np

from . import abi

DEFAULT_BAND_ROWS = 8  # one row of 8x8 wave tiles


def n_bands(height: int, band_rows: int) -> int:
    return -(-height // band_rows)


def rank_params(base, rank: int, world: int, band_rows: int = DEFAULT_BAND_ROWS):
    """rt_params of the strip rank `rank` of `world` renders of the frame in `base`."""
    p = np.array(base, dtype=abi.PARAMS_DTYPE)
    nb = n_bands(int(p["height"]), band_rows)
    p["band_rows"] = band_rows
    p["band_first"] = rank
    p["band_step"] = world
    p["band_count"] = len(range(rank, nb, world))
    return p


def max_strip_rows(base, world: int, band_rows: int = DEFAULT_BAND_ROWS) -> int:
    """Rows of the largest strip (rank 0's): gather buffers are padded to this."""
    return abi.strip_rows(rank_params(base, 0, world, band_rows))


def strip_row_index(base, world: int, band_rows: int = DEFAULT_BAND_ROWS):
    """For the concatenation of all ranks' PADDED strips (world * max_rows rows), the image
    row each padded row holds, -1 for padding.  Used once to build the de-interleave map."""
    mr = max_strip_rows(base, world, band_rows)
    idx = np.full(world * mr, -1, dtype=np.int64)
    for r in range(world):
        rows = abi.strip_image_rows(rank_params(base, r, world, band_rows))
        idx[r * mr : r * mr + len(rows)] = rows
    return idx


def frame_gather_map(base, world: int, band_rows: int = DEFAULT_BAND_ROWS):
    """src[y] = padded-strip row that holds image row y, so frame = gathered[src]."""
    idx = strip_row_index(base, world, band_rows)
    h = int(np.asarray(base)["height"])
    src = np.full(h, -1, dtype=np.int64)
    valid = idx >= 0
    src[idx[valid]] = np.nonzero(valid)[0]
    assert (src >= 0).all(), "partition does not cover the frame"
    return src


def assemble_frame(gathered: np.ndarray, base, world: int, band_rows: int = DEFAULT_BAND_ROWS) -> np.ndarray:
    """gathered: [world, max_rows, width, C] padded strips in rank order -> [height, width, C]."""
    src = frame_gather_map(base, world, band_rows)
    flat = gathered.reshape((-1,) + gathered.shape[2:])
    return flat[src]


def batch_frames(n_frames: int, rank: int, world: int):
    """Frames of a multi-frame batch rendered by `rank`: f mod world == rank, no collective."""
    return list(range(rank, n_frames, world))


class StripGatherer:
    """One frame per call, split over the ranks of a torch.distributed group and gathered
    to rank 0 (backend nccl = RCCL on the GPUs, gloo on CPU ranks).

    render_strip(params, out) must fill `out` ([rows, width, 3] uint8 torch tensor on
    `device`, a view of the padded send buffer) with the strip described by `params`,
    enqueued on the current stream.  Two send buffers alternate so that the gather of
    frame f overlaps the render of frame f+1.
    """

    def __init__(self, base, render_strip, device, group=None, band_rows: int = DEFAULT_BAND_ROWS):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world = dist.get_world_size(group)
        else:  # single process: the whole frame, no collective
            self.rank, self.world = 0, 1
        self.base = np.array(base, dtype=abi.PARAMS_DTYPE)
        self.band_rows = band_rows
        self.params = rank_params(self.base, self.rank, self.world, band_rows)
        self.rows = abi.strip_rows(self.params)
        self.max_rows = max_strip_rows(self.base, self.world, band_rows)
        self.width = int(self.base["width"])
        self.render_strip = render_strip
        self.device = device
        shape = (self.max_rows, self.width, 3)
        self.send = [torch.zeros(shape, dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = None
        self.src = None
        if self.rank == 0:
            self.recv = [
                [torch.zeros(shape, dtype=torch.uint8, device=device) for _ in range(self.world)] for _ in range(2)
            ]
            self.src = torch.from_numpy(frame_gather_map(self.base, self.world, band_rows)).to(device)
        self.pending = [None, None]
        self.turn = 0
        # see StripExchange.staged
        self.staged = self.world > 1 and dist.get_backend(group) == "gloo" and torch.device(device).type != "cpu"

    def submit(self):
        """Render this rank's strip of the next frame and start its gather.  Returns the
        buffer slot; the frame is complete on rank 0 after wait(slot)."""
        s = self.turn
        self.turn ^= 1
        if self.pending[s] is not None:
            self.pending[s].wait()
            self.pending[s] = None
        self.render_strip(self.params, self.send[s][: self.rows])
        if self.world == 1:
            return s
        if self.staged:
            host = [self.torch.empty(self.send[s].shape, dtype=self.torch.uint8) for _ in range(self.world)]
            self.dist.gather(self.send[s].cpu(), host if self.rank == 0 else None, dst=0, group=self.group)
            if self.rank == 0:
                for dst, src in zip(self.recv[s], host):
                    dst.copy_(src)
            return s
        self.pending[s] = self.dist.gather(
            self.send[s], self.recv[s] if self.rank == 0 else None, dst=0, group=self.group, async_op=True
        )
        return s

    def wait(self, slot: int):
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None

    def frame(self, slot: int):
        """Rank 0: the assembled [height, width, 3] frame of `slot` (after wait)."""
        assert self.rank == 0
        self.wait(slot)
        if self.world == 1:
            return self.send[slot][: self.rows]
        flat = self.torch.stack(self.recv[slot]).reshape(-1, self.width, 3)
        return flat.index_select(0, self.src)

    def drain(self):
        for s in (0, 1):
            self.wait(s)


class StripExchange:
    """A batch of `world` frames per call: frame slot j is owned by rank j, every rank
    renders its strip of every frame of the batch, and ONE all-to-all (RCCL over xGMI on
    the GPUs, gloo on CPU ranks) moves strip (rank r, frame j) to rank j.  This is the
    gather above run for `world` frames at once with a different root per frame, so every
    xGMI link and every GPU's HBM carries the same share and no rank is the bottleneck.
    Work per GPU and per call is one frame's worth of pixels whatever the world size.

    render_strips(slot, params, outs) must fill outs[j] ([rows, width, 3] uint8 torch tensor
    on `device`) with this rank's strip (described by `params`) of frame j of the current
    batch, for every j, enqueued on the current stream (on the GPUs: one batched launch,
    rt_render_device_batch).  `slot` (0 or 1) tells which of the two alternating buffer
    sets `outs` belongs to, so that a caller can prebuild its launch arguments per set:
    the exchange of batch b overlaps the render of batch b+1.
    """

    def __init__(self, base, render_strips, device, group=None, band_rows: int = DEFAULT_BAND_ROWS,
                 exchange_alone: bool = False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world = dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        # exchange_alone: run the collective even in a group of one rank (a copy), so that
        # the whole call sequence can be exercised on a single GPU
        self.collective = self.world > 1 or (exchange_alone and dist.is_available() and dist.is_initialized())
        self.base = np.array(base, dtype=abi.PARAMS_DTYPE)
        self.params = rank_params(self.base, self.rank, self.world, band_rows)
        self.rows = abi.strip_rows(self.params)
        self.max_rows = max_strip_rows(self.base, self.world, band_rows)
        self.width = int(self.base["width"])
        self.render_strips = render_strips
        shape = (self.world, self.max_rows, self.width, 3)
        self.send = [torch.zeros(shape, dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [torch.zeros(shape, dtype=torch.uint8, device=device) for _ in range(2)] if self.collective else None
        self.outs = [[self.send[s][j, : self.rows] for j in range(self.world)] for s in range(2)]
        self.src = torch.from_numpy(frame_gather_map(self.base, self.world, band_rows)).to(device)
        self.pending = [None, None]
        self.turn = 0
        # gloo cannot exchange device tensors: stage through the host (rehearsals of the
        # multi-rank code on a box without one GPU per rank; never the measured path)
        self.staged = self.collective and dist.get_backend(group) == "gloo" and torch.device(device).type != "cpu"

    def _exchange_staged(self, s):
        host_send = self.send[s].cpu()
        host_recv = self.torch.empty_like(host_send)
        self.dist.all_to_all_single(host_recv, host_send, group=self.group)
        self.recv[s].copy_(host_recv)

    def submit(self):
        """Render this rank's strips of the next batch and start the exchange.  Returns
        the buffer slot; this rank's frame of the batch is complete after wait(slot)."""
        s = self.turn
        self.turn ^= 1
        self.wait(s)
        self.render_strips(s, self.params, self.outs[s])
        if self.staged:
            self._exchange_staged(s)
        elif self.collective:
            self.pending[s] = self.dist.all_to_all_single(self.recv[s], self.send[s], group=self.group, async_op=True)
        return s

    def wait(self, slot: int):
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None

    def frame(self, slot: int):
        """The assembled [height, width, 3] frame this rank owns in the batch of `slot`."""
        self.wait(slot)
        buf = self.recv[slot] if self.collective else self.send[slot]
        return buf.reshape(-1, self.width, 3).index_select(0, self.src)

    def drain(self):
        for s in (0, 1):
            self.wait(s)

