"""Host-side logic of the BVH builder (no GPU): hr_bvh_build_info builds what hr_scene_create would upload.

The traversal keeps one stack entry per BVH level (64 entries: 16 in LDS + 48 private), so the builder caps its depth: SAH
(object / spatial) splits down to binary depth 36, object-median splits below (csrc/bvh_build.cpp kSahDepth).  A chain of nested slivers of
geometrically growing size makes the SAH peel a few triangles per level — the adversarial input ADVICE r1 / VERDICT r1 #9 name."""
import numpy as np
import pytest

from hybrid_rendering_amd import api, synth


def nested_sliver_chain(n=250, ratio=2.0):
    """triangle i: a sliver of size ratio^i around the origin, centroid at ratio^i along x (all boxes nested)"""
    s = (ratio ** (np.arange(n) - n // 2)).astype(np.float64)
    v = np.zeros((n, 3, 3), np.float64)
    v[:, 0] = np.stack([s * 0.9, -s * 0.5, -s * 0.01], 1)
    v[:, 1] = np.stack([s * 1.1, s * 0.5, s * 0.01], 1)
    v[:, 2] = np.stack([s * 1.0, s * 0.5, s * 0.02], 1)
    return v.astype(np.float32)


def test_adversarial_chain_stays_below_the_traversal_stack(monkeypatch):
    v = nested_sliver_chain()
    info = api.bvh_build_info(v)
    assert info.n_tris == 250 and 0 < info.max_depth < 64
    # without the SAH the same input gives a balanced tree: depth <= ceil(log2 n)
    monkeypatch.setenv("HR_BVH_SAH_DEPTH", "0")
    median = api.bvh_build_info(v)
    assert median.max_depth <= 8
    assert info.tri_bytes == median.tri_bytes == 250 * 48


def test_depth_cap_bounds_every_scene(monkeypatch):
    """median splits from level k on: depth <= k + ceil(log2 n) whatever the geometry"""
    sd = synth.sponza_like(0.25)
    base = api.bvh_build_info(sd.verts)
    assert base.max_depth < 32
    # spatial splits reference a triangle from more than one leaf, within the duplication budget (30 %)
    assert sd.n_tris * 48 <= base.tri_bytes <= int(sd.n_tris * 1.3) * 48
    for k in (0, 3):
        monkeypatch.setenv("HR_BVH_SAH_DEPTH", str(k))
        info = api.bvh_build_info(sd.verts)
        n_refs = info.tri_bytes // 48
        assert info.max_depth <= k + int(np.ceil(np.log2(n_refs)))
        assert sd.n_tris * 48 <= info.tri_bytes <= base.tri_bytes
        if k == 0:
            assert info.tri_bytes == sd.n_tris * 48   # median splits only: one reference per triangle
    monkeypatch.delenv("HR_BVH_SAH_DEPTH")
    monkeypatch.setenv("HR_BVH_SBVH", "0")
    assert api.bvh_build_info(sd.verts).tri_bytes == sd.n_tris * 48


def _big_and_small(seed=3, n_small=4000):
    """what spatial splits are for: wall-sized triangles (diagonal slabs through the whole box) over a cloud of small ones"""
    rng = np.random.RandomState(seed)
    c = rng.uniform(-100, 100, (n_small, 1, 3))
    small = c + rng.normal(size=(n_small, 3, 3)) * 1.5
    big = np.array([[[-100, -100, -100], [100, -100, 100], [100, 100, 100]], [[-100, -100, -100], [100, 100, 100], [-100, 100, -100]],
                    [[-100, 100, 100], [100, -100, -100], [100, 100, -100]], [[-120, 0, -120], [120, 0, -120], [120, 0.5, 120]]], np.float64)
    return np.concatenate([small, big]).astype(np.float32)


@pytest.mark.parametrize("env", [{}, {"HR_BVH_SBVH": "0"}, {"HR_BVH_REINSERT": "0"}, {"HR_BVH_ALPHA": "1e-7", "HR_BVH_BUDGET": "1.0"},
                                 {"HR_BVH_SPLIT": "0.05"}, {"HR_BVH_GREEDY": "1"}, {"HR_BVH_SAH_DEPTH": "2"}])
def test_every_point_of_every_triangle_is_found(monkeypatch, env):
    """hr_bvh_selfcheck: from corners, edge midpoints, centroid and random interior points of every triangle the tree must lead to
    a leaf holding that triangle — with spatial splits the references of a triangle each cover a clipped part of it."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for verts in (_big_and_small(), synth.sponza_like(0.2).verts, synth.cornell32().verts, nested_sliver_chain(60)):
        assert api.bvh_selfcheck(verts, 16) == 0


def test_spatial_splits_happen_and_stay_within_the_budget(monkeypatch):
    v = _big_and_small()
    monkeypatch.setenv("HR_BVH_SBVH", "0")
    plain = api.bvh_build_info(v)
    monkeypatch.delenv("HR_BVH_SBVH")
    sbvh = api.bvh_build_info(v)
    assert plain.tri_bytes == len(v) * 48
    assert len(v) * 48 < sbvh.tri_bytes <= int(len(v) * 1.3) * 48
    monkeypatch.setenv("HR_BVH_BUDGET", "0.01")
    assert api.bvh_build_info(v).tri_bytes <= int(len(v) * 1.01) * 48


def test_build_is_deterministic(monkeypatch):
    """same input -> same tree, whatever the number of builder threads (subtrees above 16k references are built concurrently and merged in
    left-right order; the duplication budget is split by reference count, not consumed in build order)"""
    v = _big_and_small(seed=9, n_small=1500)
    a, b = api.bvh_build_info(v), api.bvh_build_info(v)
    assert (a.n_nodes, a.max_depth, a.tri_bytes, a.node_bytes) == (b.n_nodes, b.max_depth, b.tri_bytes, b.node_bytes)
    big = synth.sponza_like(0.6).verts      # ~100k triangles: several parallel levels
    shapes = []
    for t in ("1", "2", "7"):
        monkeypatch.setenv("HR_BVH_THREADS", t)
        i = api.bvh_build_info(big)
        shapes.append((i.n_nodes, i.max_depth, i.tri_bytes, i.node_bytes))
    assert shapes[0] == shapes[1] == shapes[2], shapes
    monkeypatch.setenv("HR_BVH_THREADS", "5")
    assert api.bvh_selfcheck(big, 8) == 0


def test_non_finite_triangles_get_no_reference(monkeypatch):
    """a triangle with a NaN / infinite coordinate can never be hit (the watertight test fails on NaN): the builder leaves it out instead of
    letting its box poison the SAH areas of every node above it; the scene bounds ignore it too"""
    import time
    rng = np.random.RandomState(0)
    v = (rng.normal(size=(3000, 1, 3)) * 50 + rng.normal(size=(3000, 3, 3))).astype(np.float32)
    w = v.copy()
    w[5, 1, 2] = np.nan; w[77, 0, 0] = np.inf; w[200] = -np.inf
    monkeypatch.setenv("HR_BVH_SBVH", "0")
    t0 = time.time()
    info, clean = api.bvh_build_info(w), api.bvh_build_info(np.delete(v, [5, 77, 200], axis=0))
    assert time.time() - t0 < 5.0
    assert info.tri_bytes == (3000 - 3) * 48 == clean.tri_bytes and info.n_nodes == clean.n_nodes
    assert np.isfinite(list(info.bounds_lo) + list(info.bounds_hi)).all() and list(info.bounds_lo) == list(clean.bounds_lo)
    nothing = api.bvh_build_info(np.full((4, 3, 3), np.nan, np.float32))
    assert nothing.n_nodes == 1 and nothing.tri_bytes == 0


def test_empty_and_single_triangle():
    assert api.bvh_build_info(np.zeros((0, 3, 3), np.float32)).n_nodes == 1
    one = api.bvh_build_info(np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32))
    assert one.n_nodes == 1 and one.tri_bytes == 48


def test_bad_material_index_is_a_status_code():
    """tri_material is validated on the host before anything is uploaded (the hit shading indexes materials[] with it)"""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs no GPU; on a GPU box tests/test_gpu_trace.py covers it through a context")
    L = api.lib()
    d = api.hr_scene_desc()
    h = C.c_void_p()
    assert L.hr_scene_create(None, C.byref(d), C.byref(h)) == 1   # HR_ERR_INVALID_ARG (no context), never an exception
