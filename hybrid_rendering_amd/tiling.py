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


def band_rows(height: int, world: int, rank: int, align: int = 8, bounds: Sequence[int] = None) -> Tuple[int, int]:
    """Contiguous row band of ``rank``; boundaries are multiples of ``align`` (8-row tiles, 4-row masks).
    ``bounds`` (world + 1 ascending row boundaries, e.g. from balanced_bounds) replaces the uniform partition."""
    if bounds is not None:
        if len(bounds) != world + 1 or bounds[0] != 0 or bounds[-1] != height or any(b % align for b in bounds[:-1]) \
                or any(b1 <= b0 for b0, b1 in zip(bounds, bounds[1:])):
            raise ValueError(f"bad band boundaries {list(bounds)} for height {height}, world {world}")
        return int(bounds[rank]), int(bounds[rank + 1])
    tiles = (height + align - 1) // align
    t0 = (tiles * rank) // world
    t1 = (tiles * (rank + 1)) // world
    return t0 * align, min(t1 * align, height)


def balanced_bounds(cost_per_tile_row: Sequence[float], world: int, height: int, min_tiles: int = 8, align: int = 8) -> List[int]:
    """Row boundaries that give every band (about) the same share of ``cost_per_tile_row`` (one entry per ``align`` rows,
    e.g. a * pixels + b * rays of a calibration frame): screen-space cost is far from uniform — sky rows fire no ray, the
    floor fires one per pixel — and the slowest band sets the frame rate.  Every band keeps at least ``min_tiles`` tile
    rows so that the halo exchange only ever involves direct neighbours."""
    c = [max(float(x), 0.0) for x in cost_per_tile_row]
    n = len(c)
    if n < world * min_tiles:
        raise ValueError("image too small for that many bands")
    total = sum(c) or 1.0
    cuts, acc, k = [0], 0.0, 1
    for i in range(n):
        acc += c[i]
        # cut after tile row i when the running cost reaches the k-th share, leaving room for the remaining bands
        while k < world and acc >= total * k / world and (i + 1) - cuts[-1] >= min_tiles:
            if n - (i + 1) < (world - k) * min_tiles:
                break
            cuts.append(i + 1)
            k += 1
    while len(cuts) < world:  # not enough cost left: give the remaining bands the minimum height from the bottom
        cuts.append(0)
    cuts = sorted(cuts[:world])
    for j in range(1, world):  # enforce min height going down, then going up
        cuts[j] = max(cuts[j], cuts[j - 1] + min_tiles)
    limit = n
    for j in range(world - 1, 0, -1):
        cuts[j] = min(cuts[j], limit - min_tiles)
        limit = cuts[j]
    return [c_ * align for c_ in cuts] + [height]



def rebalanced_bounds(bounds: Sequence[int], times: Sequence[float], height: int, align: int = 16, min_rows: int = 64, fixed_fraction: float = 0.35) -> List[int]:
    """New band boundaries from MEASURED per-rank frame times (one per band of ``bounds``).  A rank's time is modelled as a fixed
    part (launch floors, its probe slab: ``fixed_fraction`` of the mean time) plus a cost density, constant over its rows; the new
    cuts give every band the same share of the integrated density.  Deterministic, so every rank computes the same answer from
    the all-gathered times.  Cuts are multiples of ``align`` rows and every band keeps ``min_rows``."""
    world = len(times)
    assert len(bounds) == world + 1
    mean = sum(times) / world
    fixed = fixed_fraction * mean
    dens = [max(times[r] - fixed, 0.05 * mean) / max(bounds[r + 1] - bounds[r], 1) for r in range(world)]   # cost per row of each old band
    n = (height + align - 1) // align
    cost = []
    for i in range(n):
        y = min(i * align + align // 2, height - 1)
        r = max(k for k in range(world) if bounds[k] <= y)
        cost.append(dens[min(r, world - 1)] * align)
    return balanced_bounds(cost, world, height, min_tiles=max(1, min_rows // align), align=align)

RAY_WEIGHT = 1.8    # cost of one shadow ray in units of one geometry pixel of denoising.  The 1080p stage times give 1.25 (trace
                    # 0.131 ns/ray, temporal + a-trous 0.105 ns/pixel); 1.8 balances the measured band times of the 4- and 8-band
                    # frames best (tools/band_balance.py: slowest/mean band 1.04 instead of 1.10) — ray-dense rows are also the deep ones
SKY_WEIGHT = 0.3    # sky pixels leave the denoise kernels early


def shadow_cost_per_tile_row(depth, tile_rays, ray_weight: float = None, sky_weight: float = SKY_WEIGHT):
    """Cost model of the shadows pass per 8-row tile row, from a calibration frame: ``depth`` = [H, W] torch depth image
    (sky == 1.0), ``tile_rays`` = [tiles_y, tiles_x] rays per tile (RayTracedShadows.tile_ray_counts)."""
    import numpy as np
    import os
    rw = float(os.environ.get("HR_RAY_WEIGHT", RAY_WEIGHT)) if ray_weight is None else ray_weight
    H, W = depth.shape
    geom = (depth != 1.0).sum(dim=1).double().cpu().numpy()
    geom_rows = np.add.reduceat(geom, np.arange(0, H, 8))
    px_rows = np.add.reduceat(np.full(H, float(W)), np.arange(0, H, 8))
    return geom_rows + sky_weight * (px_rows - geom_rows) + rw * np.asarray(tile_rays, np.float64).sum(axis=1)


def exchange_plan(height: int, world: int, rank: int, rows: int, bounds: Sequence[int] = None):
    """Row ranges for one neighbour exchange.  Returns a list of (peer, send_rows, recv_rows) with absolute
    half-open row ranges: I send rows of MY band adjacent to the peer and receive the peer's adjacent rows."""
    b0, b1 = band_rows(height, world, rank, bounds=bounds)
    plan = []
    if rank > 0:
        p0, p1 = band_rows(height, world, rank - 1, bounds=bounds)
        plan.append((rank - 1, (b0, min(b0 + rows, b1)), (max(p1 - rows, p0), p1)))
    if rank < world - 1:
        p0, p1 = band_rows(height, world, rank + 1, bounds=bounds)
        plan.append((rank + 1, (max(b1 - rows, b0), b1), (p0, min(p0 + rows, p1))))
    return plan


def _host_backend_fence(tensors, group=None):
    """RCCL orders its transfers after the kernels already enqueued on the current stream.  gloo (only used to exercise this
    code on CPUs, or with several ranks on one GPU) reads device memory from the host at call time instead: wait for the
    producers first."""
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) != "nccl" and any(getattr(t, "is_cuda", False) for t in tensors):
        torch.cuda.synchronize()


# the collective this rank posted last, and how many it has posted (bench.py's watchdog prints it when a run hangs on a real multi-GPU node)
LAST_COLLECTIVE = {"name": None, "count": 0}


def _note_collective(name: str):
    LAST_COLLECTIVE["name"] = name
    LAST_COLLECTIVE["count"] += 1


def exchange_halo(images: Sequence, height: int, world: int, rank: int, rows: int, group=None, wait: bool = True, bounds: Sequence[int] = None):
    """Grouped neighbour exchange of ``rows`` halo rows for every [H, W, ...] tensor in ``images`` (all ranks
    pass the same list in the same order).  Works on any backend (nccl on GPUs, gloo in the CPU tests).
    ``wait=False`` returns the pending requests (+ the tensors they pin) instead of waiting: the caller waits right
    before the first kernel that reads the received rows, so the transfer overlaps with whatever is enqueued earlier."""
    import torch.distributed as dist
    if world == 1:
        return []
    _host_backend_fence(images, group)
    ops, recvs = [], []
    for peer, (s0, s1), (r0, r1) in exchange_plan(height, world, rank, rows, bounds):
        for img in images:
            send = img[s0:s1].contiguous()
            recv = img[r0:r1]
            if not recv.is_contiguous():
                raise ValueError("halo rows must be a contiguous slab (row-major images)")
            ops.append(dist.P2POp(dist.isend, send, peer, group))
            ops.append(dist.P2POp(dist.irecv, recv, peer, group))
            recvs.append(send)  # keep alive until the batch completes
    _note_collective(f"exchange_halo({len(images)} images x {rows} rows, rank {rank}/{world})")
    reqs = dist.batch_isend_irecv(ops)
    if not wait:
        return [(reqs, recvs)]
    for req in reqs:
        req.wait()
    return []


class _TiledPass:
    """A denoised pass on one band of a row-tiled frame: ``render()`` = the reference's render() for this band + the
    per-frame exchange of the history rows next to the band boundaries."""

    def __init__(self, rank: int, world: int, height: int, history_rows: int, group=None, bounds: Sequence[int] = None):
        self.rank, self.world, self.height, self.group = rank, world, height, group
        self.bounds = list(bounds) if bounds is not None else None
        self.b0, self.b1 = band_rows(height, world, rank, bounds=self.bounds)
        self.history_rows = history_rows
        # the exchange only talks to the two direct neighbours: a neighbour band shorter than the history apron would leave
        # apron rows owned by the SECOND neighbour unrefreshed (stale history => bands silently differ from one GPU)
        if world > 1:
            short = [r for r in range(world) if (lambda b: b[1] - b[0])(band_rows(height, world, r, bounds=self.bounds)) < history_rows]
            if short:
                raise ValueError(f"bands {short} of a {height}-row image cut {world} ways are shorter than the {history_rows}-row history apron; "
                                 "use fewer bands (or pass bounds= with taller bands)")

    def history_images(self, ping_pong: int) -> List:
        raise NotImplementedError

    _pending: list = []

    def _render(self, scene, inputs, stream, *extra):
        self.wait_exchange()
        self.pass_.render(scene, inputs, *extra, stream=stream)

    def time_exchange(self, n: int = 20) -> float:
        """microseconds per neighbour exchange of this pass's history rows, posted and waited for back to back on the current stream
        (HIP events around n repetitions, one synchronisation): the communication cost of a frame that the overlap with the next
        frame's ray trace hides — reported by bench.py next to the frame time so that compute and communication can be told apart"""
        import torch
        if self.world == 1:
            return 0.0
        self.wait_exchange()
        imgs = self.history_images(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(n + 2):
            if k == 2:
                e0.record()
            exchange_halo(imgs, self.height, self.world, self.rank, self.history_rows, self.group, wait=True, bounds=self.bounds)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    def history_apron_exceeded(self) -> bool:
        """Runtime guard of the history apron: True if, since the last call, a history tap of this band fell on an image row this GPU
        does not hold (per-frame motion beyond ``history_rows`` minus the halo).  Such taps read as disoccluded, so the band is still a
        valid image but no longer identical to the single-GPU one; the caller widens the apron.  Synchronises the device."""
        return self.pass_.history_apron_exceeded()

    def wait_exchange(self):
        """Orders everything enqueued afterwards behind the previous frame's halo exchange.  With RCCL req.wait() only
        makes the current stream wait for the communication stream (no host synchronisation); with gloo it blocks."""
        for reqs, _keep in self._pending:
            for req in reqs:
                req.wait()
        self._pending = []

    def render(self, scene, inputs, *extra, stream=None):
        """``stream``: torch stream to record on (default: the current one).  torch.distributed orders its RCCL operations and
        req.wait() against torch's CURRENT stream, so a caller-supplied stream is made current for the whole frame — kernels,
        exchange and waits then share one ordering domain."""
        import torch
        if stream is not None and torch.cuda.is_available() and stream != torch.cuda.current_stream():
            with torch.cuda.stream(stream):
                return self.render(scene, inputs, *extra, stream=None)
        self._render(scene, inputs, stream, *extra)
        if self.world > 1:
            # torch.distributed orders the NCCL/RCCL ops after the kernels already enqueued on the current stream.  The
            # received rows are history for the NEXT frame's temporal pass only, so nobody waits here: the next frame's
            # ray-trace kernel (which reads no history) runs while the rows are in flight; _render() waits before the
            # first reader.
            self._pending = exchange_halo(self.history_images(int(inputs.ping_pong)), self.height, self.world, self.rank, self.history_rows,
                                          self.group, wait=False, bounds=self.bounds)


class TiledShadows(_TiledPass):
    """RayTracedShadows on one band (halo 24 = 8 mask rows + 15 a-trous rows; 40 history rows)."""

    def __init__(self, ctx, width: int, height: int, rank: int, world: int, halo: int = HALO, history_halo: int = HISTORY_HALO, group=None,
                 bounds: Sequence[int] = None):
        from . import api
        super().__init__(rank, world, height, history_halo, group, bounds)
        self.history_halo = history_halo
        band = (self.b0, self.b1, halo, history_halo) if world > 1 else None
        self.pass_ = api.RayTracedShadows(ctx, width, height, api.SCALE_FULL_RES, band=band)
        self.params = self.pass_.params

    def history_images(self, ping_pong: int) -> List:
        p = self.pass_
        return [p.image(p.IMG_PREV), p.image(p.IMG_MOMENTS1 if ping_pong else p.IMG_MOMENTS0)]

    def _render(self, scene, inputs, stream):
        # RayTracedShadows::render stage by stage (ray_traced_shadows.cpp:100-116): the trace needs no history, so it is
        # enqueued BEFORE waiting for the halo rows of the previous frame
        p = self.pass_
        p.ray_trace(scene, inputs, stream)
        self.wait_exchange()
        p.denoise(inputs, stream)      # temporal + a-trous chain: the launches render() makes (tolerance mode: iterations 0 + 1 fused)

    def band_output(self, kind=None):
        from . import api
        self.wait_exchange()
        out = self.pass_.output(api.OUTPUT_ATROUS if kind is None else kind)
        return out[self.b0:self.b1]


class TiledAO(_TiledPass):
    """RayTracedAO on one band of the pass image (``height`` etc. are in PASS resolution: full >> scale).  The halo
    (24 rows >= 8 mask rows + 4 blur rows + motion) is also the history apron: the temporal output and the history
    length of the ``halo`` rows next to each boundary are refreshed from their owner every frame; history taps
    further out read as disoccluded (exact while per-frame motion stays below halo - 12 rows)."""

    def __init__(self, ctx, full_width: int, full_height: int, rank: int, world: int, scale: int = 0, halo: int = HALO, group=None,
                 bounds: Sequence[int] = None):
        from . import api
        super().__init__(rank, world, full_height >> scale, halo, group, bounds)
        band = (self.b0, self.b1, halo, halo) if world > 1 else None
        self.pass_ = api.RayTracedAO(ctx, full_width, full_height, scale, band=band)
        self.params = self.pass_.params

    def history_images(self, ping_pong: int) -> List:
        p = self.pass_
        return [p.image(p.IMG_AO1 if ping_pong else p.IMG_AO0), p.image(p.IMG_LEN1 if ping_pong else p.IMG_LEN0)]

    def _render(self, scene, inputs, stream):
        self.wait_exchange()
        self.pass_.render(scene, inputs, stream)


class TiledReflections(_TiledPass):
    """RayTracedReflections on one band of the pass image (8-row colour apron of the temporal pass + 15 a-trous rows
    fit in halo = 24; history = feedback image + moments)."""

    def __init__(self, ctx, full_width: int, full_height: int, rank: int, world: int, scale: int = 0, halo: int = HALO, group=None,
                 bounds: Sequence[int] = None):
        from . import api, api_reflections
        super().__init__(rank, world, full_height >> scale, halo, group, bounds)
        band = (self.b0, self.b1, halo, halo) if world > 1 else None
        self.pass_ = api_reflections.RayTracedReflections(ctx, full_width, full_height, scale, band=band)
        self.params = self.pass_.params

    def history_images(self, ping_pong: int) -> List:
        # colour history of the NEXT frame's temporal stage: the feedback image only with blur_as_input, else this frame's temporal
        # output (ray_traced_reflections.cpp:1124,1218 — the reference's default); + the moments written this frame
        p = self.pass_
        colour = p.image(p.IMG_PREV) if int(p.params.blur_as_input) else p.image(p.IMG_COLOR1 if ping_pong else p.IMG_COLOR0)
        return [colour, p.image(p.IMG_MOMENTS1 if ping_pong else p.IMG_MOMENTS0)]

    def _render(self, scene, inputs, stream, env, ddgi):
        self.wait_exchange()
        self.pass_.render(scene, inputs, env, ddgi, stream=stream)


# ---------------------------------------------------------------------------------------------- DDGI
def probe_slabs(cz: int, world: int, rank: int) -> Tuple[int, int]:
    """z-slabs [z0, z1) of the probe grid owned by ``rank`` (every rank gets at least one when cz >= world)."""
    return (cz * rank) // world, (cz * (rank + 1)) // world


def slab_rows(side: int, z0: int, z1: int) -> Tuple[int, int]:
    """Atlas rows of probe z-slabs [z0, z1): probes are laid out x + y*cx along the atlas x axis and z along y, each
    probe (side + 2) texels wide, inside a 1-texel frame (ddgi.cpp:197-201) -> one contiguous row range."""
    return 1 + z0 * (side + 2), 1 + z1 * (side + 2)


def allgather_slabs(atlas, side: int, cz: int, world: int, rank: int, group=None):
    """Every rank has written the atlas rows of its own probe slabs; afterwards every rank holds all rows.
    Equal slabs -> one all_gather of row blocks; ragged slabs -> one broadcast per owner."""
    import torch.distributed as dist
    if world == 1:
        return
    rows = [slab_rows(side, *probe_slabs(cz, world, r)) for r in range(world)]
    _host_backend_fence([atlas], group)
    _note_collective(f"allgather_slabs(side {side}, rank {rank}/{world})")
    if len({b - a for a, b in rows}) == 1:
        outs = [atlas[a:b] for a, b in rows]
        dist.all_gather(outs, atlas[rows[rank][0]:rows[rank][1]].clone(), group=group)
    else:
        for r, (a, b) in enumerate(rows):
            if b > a:
                dist.broadcast(atlas[a:b], src=r, group=group)


class ShardedDDGI:
    """DDGI across the GPUs of a node: probes are partitioned by z-slab for the ray trace and the
    probe updates, the freshly written atlas rows are all-gathered (irradiance 1.7 MB + depth 2.7 MB in total for a
    16x8x16 grid), then every rank samples the full atlases for its own row band of the image."""

    def __init__(self, ctx, width: int, height: int, np_ddgi, rank: int, world: int, scale: int = 0, group=None):
        from . import api_gi
        self.rank, self.world, self.group = rank, world, group
        self.pass_ = api_gi.DDGI(ctx, width, height, np_ddgi, scale)
        self.params = self.pass_.params
        self.cz = int(np_ddgi["probe_counts"][2])
        self.irr_side, self.dep_side = int(np_ddgi["irradiance_probe_side_length"]), int(np_ddgi["depth_probe_side_length"])
        if self.cz < world:
            raise ValueError("ShardedDDGI needs at least one probe z-slab per rank")
        self.z0, self.z1 = probe_slabs(self.cz, world, rank)
        self.b0, self.b1 = band_rows(height >> scale, world, rank)
        if world > 1:
            self.pass_.set_shard(self.z0, self.z1, self.b0, self.b1)

    def render(self, scene, inputs, env, orientation=None, stream=None):
        p = self.pass_
        if orientation is not None:
            p.set_orientation(orientation)
        p.ray_trace(scene, inputs, env, stream)
        p.probe_update(stream)
        irr, dep = p.current_write()
        allgather_slabs(irr, self.irr_side, self.cz, self.world, self.rank, self.group)
        allgather_slabs(dep, self.dep_side, self.cz, self.world, self.rank, self.group)
        p.sample_probe_grid(inputs, stream)
        p.end_frame()

    def time_allgather(self, n: int = 20) -> float:
        """microseconds per frame for the two atlas all-gathers (irradiance + depth), HIP events on the current stream"""
        import torch
        if self.world == 1:
            return 0.0
        irr, dep = self.pass_.current_write()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(n + 2):
            if k == 2:
                e0.record()
            allgather_slabs(irr, self.irr_side, self.cz, self.world, self.rank, self.group)
            allgather_slabs(dep, self.dep_side, self.cz, self.world, self.rank, self.group)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    def band_output(self):
        return self.pass_.output()[self.b0:self.b1]
