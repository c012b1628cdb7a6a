"""Fuzzer of the BVH builder (csrc/bvh_build.cpp: spatial splits, reference unsplitting, reinsertion, collapse, quantisation) on random
triangle soups — geometry the fixed test scenes do not have: wall-sized triangles over clouds of small ones, slivers, coplanar
sheets, exact duplicates, clusters many orders of magnitude apart, flat (2-D) scenes, random build switches.

    python tools/fuzz_bvh.py [seed] [n]            CPU: hr_bvh_selfcheck — from sampled points of every triangle the tree must lead to a
                                                   leaf holding that triangle (the invariant every query relies on)
    python tools/fuzz_bvh.py [seed] [n] --gpu      GPU: any-hit and closest-hit answers of the HIP traversal against the oracle's scalar BVH2
                                                   traversal on random rays (bit-exact t, u, v, primitive)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hybrid_rendering_amd import api as hr, synth


def soup(rng):
    kind = rng.choice(["cloud", "walls", "slivers", "sheets", "dupes", "scales", "flat", "grid"])
    n = int(rng.choice([1, 2, 3, 7, 40, 300, 2500, 12000]))
    ext = float(10.0 ** rng.uniform(-2, 3))
    c = rng.uniform(-ext, ext, (n, 1, 3))
    size = ext * float(10.0 ** rng.uniform(-3, -0.5))
    v = c + rng.normal(size=(n, 3, 3)) * size
    if kind == "walls":       # a few triangles spanning the whole scene, at random orientations
        k = int(rng.randint(1, 9))
        big = rng.uniform(-ext, ext, (k, 3, 3)) * 1.2
        v = np.concatenate([v, big])
    elif kind == "slivers":   # long thin triangles
        d = rng.normal(size=(n, 1, 3)); d /= np.linalg.norm(d, axis=2, keepdims=True)
        t = np.linspace(-1, 1, 3)[None, :, None] * ext * rng.uniform(0.05, 1.0, (n, 1, 1))
        v = c + d * t + rng.normal(size=(n, 3, 3)) * size * 1e-3
    elif kind == "sheets":    # coplanar, overlapping layers
        v[:, :, int(rng.randint(3))] = np.round(v[:, :, int(rng.randint(3))] / (ext * 0.25)) * (ext * 0.25)
    elif kind == "dupes":     # exact duplicates (equal t: the tie rule decides)
        v = np.concatenate([v, v[rng.randint(0, n, max(1, n // 3))]])
    elif kind == "scales":    # two clusters many orders of magnitude apart in size
        v = np.concatenate([v, rng.normal(size=(max(1, n // 2), 3, 3)) * ext * 1e-4 + ext * 0.3])
    elif kind == "flat":      # the whole scene in one plane
        v[:, :, 1] = 0.0
    elif kind == "grid":      # regular tessellation (equal centroids along axes: SAH ties)
        g = int(max(1, np.sqrt(n / 2)))
        xs, ys = np.meshgrid(np.arange(g + 1) * ext / g, np.arange(g + 1) * ext / g, indexing="ij")
        P = np.stack([xs, np.zeros_like(xs), ys], -1)
        a, b, c2, d2 = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
        v = np.concatenate([np.stack([a, b, c2], -2).reshape(-1, 3, 3), np.stack([a, c2, d2], -2).reshape(-1, 3, 3)])
    return str(kind), np.ascontiguousarray(v, np.float32)


SWITCHES = [{}, {}, {"HR_BVH_SBVH": "0"}, {"HR_BVH_REINSERT": "0"}, {"HR_BVH_ALPHA": "1e-8", "HR_BVH_BUDGET": "2.0"}, {"HR_BVH_REINSERT": "4", "HR_BVH_REINSERT_FRACTION": "1.0", "HR_BVH_REINSERT_MAX_AREA": "1.0"},
            {"HR_BVH_SAH_DEPTH": "3"}, {"HR_BVH_GREEDY": "1"}, {"HR_BVH_SPLIT": "0.1"}, {"HR_BVH_BUDGET": "0.02"}]


def with_env(env):
    for k in list(os.environ):
        if k.startswith("HR_BVH_"):
            del os.environ[k]
    os.environ.update(env)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 50
    gpu = "--gpu" in sys.argv
    rng = np.random.RandomState(seed)
    bad = 0
    if gpu:
        import torch
        from oracle import pyoracle as oracle
        ctx = hr.Context(0)
    for t in range(n):
        kind, v = soup(rng)
        env = dict(SWITCHES[int(rng.randint(len(SWITCHES)))])
        with_env(env)
        tag = f"trial {t}: {kind} {len(v)} triangles {env}"
        if not gpu:
            info = hr.bvh_build_info(v)
            miss = hr.bvh_selfcheck(v, 14)
            ok = miss == 0 and 0 < info.max_depth < 64 and info.tri_bytes >= len(v) * 48
            if not ok:
                bad += 1
                print("FAIL", tag, "uncovered", miss, "depth", info.max_depth, flush=True)
            continue
        sd = synth.SceneData(v, np.zeros_like(v), np.zeros(len(v), np.uint32), np.ones(len(v), np.uint32), np.array([[0.5] * 3 + [0, 0.5, 0, 0, 0]], np.float32), "soup")
        osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
        lo, hi = sd.bounds()
        m = 60000
        o = rng.uniform(lo - 0.1 * (hi - lo) - 1e-3, hi + 0.1 * (hi - lo) + 1e-3, size=(m, 3))
        tgt = v.reshape(-1, 3)[rng.randint(0, len(v) * 3, m)] + rng.normal(size=(m, 3)) * 0.02 * float(np.linalg.norm(hi - lo) + 1e-6)   # aim at the geometry
        d = tgt - o
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
        d[::53] = np.eye(3)[rng.randint(0, 3, size=len(d[::53]))] * rng.choice([-1.0, 1.0], size=(len(d[::53]), 1))
        rays = np.zeros((m, 8), np.float32)
        rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 1.0e4 * max(1.0, float(np.linalg.norm(hi - lo))), d, 0.0
        rays[::2, 3] = rng.uniform(0.0, float(np.linalg.norm(hi - lo)) * 1.5 + 1e-6, size=len(rays[::2]))
        ref = osc.any_hit(rays)
        got = gsc.any_hit(torch.from_numpy(rays).cuda()).cpu().numpy()
        tuv, prim = osc.closest_hit(rays)
        gt, gp = gsc.closest_hit(torch.from_numpy(rays).cuda())
        gt, gp = gt.cpu().numpy(), gp.cpu().numpy()
        hit = prim >= 0
        ok = np.array_equal(ref != 0, got != 0) and np.array_equal(prim, gp) and np.array_equal(tuv[hit].view(np.uint32), gt[hit].view(np.uint32))
        if not ok:
            bad += 1
            print("FAIL", tag, "any-hit mismatches", int(((ref != 0) != (got != 0)).sum()), "closest prim mismatches", int((prim != gp).sum()), flush=True)
        gsc.close()
    print(f"fuzz_bvh seed {seed}: {n} soups ({'GPU queries vs oracle' if gpu else 'CPU coverage self-check'}), {bad} failures", flush=True)
    sys.exit(1 if bad else 0)


main()
