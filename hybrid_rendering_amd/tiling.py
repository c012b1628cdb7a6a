"""Row tiling of the ray-trace + denoise path across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The frame is cut into
contiguous row bands; BVH, blue-noise tables, UBO and the G-buffer rows a band can reach are replicated.

Per frame and band:
  * trace + temporal + the whole a-trous chain are computed on the band plus ``halo`` rows on each side
    (zero-communication redundant compute: the 1+2+4+8 = 15-row footprint of the chain plus the 8-row mask
    footprint of the temporal pass fit in halo = 24 rows; rows closer than that to the halo edge are exact);
  * ONE grouped neighbour exchange (send/recv up + send/recv down, 2 of the 7 xGMI links) refreshes the
    ``history_halo`` rows of the two images the NEXT frame reprojects from — the a-trous feedback image and
    the moments — with the owner's exact values.  No all-reduce anywhere.

With history_halo >= halo + (largest per-frame motion in rows) every band row is bit-identical to the
single-GPU result (tests/test_gpu_tiling.py).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

HALO = 24          # multiple of 8 >= 8 (mask footprint) + 15 (a-trous chain)
HISTORY_HALO = 40  # HALO + 16 rows of motion


def band_rows(height: int, world: int, rank: int, align: int = 8) -> Tuple[int, int]:
    """Contiguous row band of ``rank``; boundaries are multiples of ``align`` (8-row tiles, 4-row masks)."""
    tiles = (height + align - 1) // align
    t0 = (tiles * rank) // world
    t1 = (tiles * (rank + 1)) // world
    return t0 * align, min(t1 * align, height)


def exchange_plan(height: int, world: int, rank: int, rows: int):
    """Row ranges for one neighbour exchange.  Returns a list of (peer, send_rows, recv_rows) with absolute
    half-open row ranges: I send rows of MY band adjacent to the peer and receive the peer's adjacent rows."""
    b0, b1 = band_rows(height, world, rank)
    plan = []
    if rank > 0:
        p0, p1 = band_rows(height, world, rank - 1)
        plan.append((rank - 1, (b0, min(b0 + rows, b1)), (max(p1 - rows, p0), p1)))
    if rank < world - 1:
        p0, p1 = band_rows(height, world, rank + 1)
        plan.append((rank + 1, (max(b1 - rows, b0), b1), (p0, min(p0 + rows, p1))))
    return plan


def exchange_halo(images: Sequence, height: int, world: int, rank: int, rows: int, group=None):
    """Grouped neighbour exchange of ``rows`` halo rows for every [H, W, ...] tensor in ``images`` (all ranks
    pass the same list in the same order).  Works on any backend (nccl on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    if world == 1:
        return
    ops, recvs = [], []
    for peer, (s0, s1), (r0, r1) in exchange_plan(height, world, rank, rows):
        for img in images:
            send = img[s0:s1].contiguous()
            recv = img[r0:r1]
            if not recv.is_contiguous():
                raise ValueError("halo rows must be a contiguous slab (row-major images)")
            ops.append(dist.P2POp(dist.isend, send, peer, group))
            ops.append(dist.P2POp(dist.irecv, recv, peer, group))
            recvs.append(send)  # keep alive until the batch completes
    for req in dist.batch_isend_irecv(ops):
        req.wait()


class TiledShadows:
    """RayTracedShadows on one band of a row-tiled frame.  ``render()`` = the reference's render() for this
    band + the per-frame history halo exchange."""

    def __init__(self, ctx, width: int, height: int, rank: int, world: int, halo: int = HALO, history_halo: int = HISTORY_HALO, group=None):
        from . import api
        self.rank, self.world, self.height, self.group = rank, world, height, group
        self.b0, self.b1 = band_rows(height, world, rank)
        self.history_halo = history_halo
        band = (self.b0, self.b1, halo, history_halo) if world > 1 else None
        self.pass_ = api.RayTracedShadows(ctx, width, height, api.SCALE_FULL_RES, band=band)
        self.params = self.pass_.params

    def history_images(self, ping_pong: int) -> List:
        p = self.pass_
        return [p.image(p.IMG_PREV), p.image(p.IMG_MOMENTS1 if ping_pong else p.IMG_MOMENTS0)]

    def render(self, scene, inputs, stream=None):
        self.pass_.render(scene, inputs, stream)
        if self.world > 1:
            # torch.distributed orders the NCCL/RCCL ops after the kernels already enqueued on the current stream
            # and req.wait() makes the current stream wait for them: no host synchronisation.
            exchange_halo(self.history_images(int(inputs.ping_pong)), self.height, self.world, self.rank, self.history_halo, self.group)

    def band_output(self, kind=None):
        from . import api
        out = self.pass_.output(api.OUTPUT_ATROUS if kind is None else kind)
        return out[self.b0:self.b1]
