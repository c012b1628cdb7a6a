"""Full-size checks at BASELINE.json's sizes (1920x1080, the ~278k-triangle Sponza-like scene, 16x8x16x256 DDGI):
bit-exact parity with the oracle where the oracle finishes in seconds, and size-independent properties everywhere
else — two differently shaped BVHs and two different trace schedulers must produce the same masks, a row-banded frame
must equal the whole frame, sharded DDGI must equal the unsharded one, a run must be reproducible."""
import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env, tiling

pytestmark = pytest.mark.gpu
W, H = 1920, 1080


@pytest.fixture(scope="module")
def full(hr, ctx):
    import torch
    sd = helpers.scene_data("sponza")
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(4)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(3)]
    gbs = [scene.gbuffer(u, W, H) for u in ubos]
    sob, sr = synth.blue_noise_tables()
    return dict(sd=sd, scene=scene, ubos=ubos, gbs=gbs, sob=sob, sr=sr, sob_d=torch.from_numpy(sob).cuda(), sr_d=torch.from_numpy(sr).cuda())


def _host(gb):
    import torch
    return {n: (t.cpu().numpy().view(np.uint16) if t.dtype == torch.float16 else t.cpu().numpy()) for n, t in gb.items()}


def _fi(hr, full, k):
    return hr.frame_inputs(full["gbs"][k], full["gbs"][k - 1 if k else 0], full["ubos"][k], k, k & 1, full["sob_d"], full["sr_d"])


def test_shadows_1080p_matches_oracle(oracle, hr, ctx, full):
    """the bench workload itself, two frames (trace + temporal + 4 a-trous), every image bit for bit (the tolerance mode on the same
    workload: test_gpu_tolerance.py::test_1080p_bench_frame_tolerance)"""
    import torch
    osc = oracle.Scene(full["sd"])
    op, gp = oracle.ShadowsPass(W, H), hr.RayTracedShadows(ctx, W, H)
    host = [_host(g) for g in full["gbs"][:2]]
    for k in range(2):
        op.render(osc, full["ubos"][k], host[k], host[k - 1 if k else 0], full["sob"], full["sr"], k)
        gp.render(full["scene"], _fi(hr, full, k))
        torch.cuda.synchronize()
        st = op.stages
        assert np.array_equal(gp.image(gp.IMG_MASK).cpu().numpy().view(np.uint32), st["mask"]), f"frame {k}: mask"
        assert gp.ray_count() == st["rays"] and st["rays"] > 500_000
        assert np.array_equal(helpers.bits16(gp.image(gp.IMG_TEMPORAL)), st["temporal"]), f"frame {k}: temporal"
        assert np.array_equal(helpers.bits16(gp.output(hr.OUTPUT_ATROUS)), st["output"]), f"frame {k}: a-trous output"
    gp.close()


def test_masks_do_not_depend_on_bvh_shape_or_scheduler(hr, ctx, full, monkeypatch):
    import torch
    ref = hr.RayTracedShadows(ctx, W, H)
    ref.ray_trace(full["scene"], _fi(hr, full, 1))
    torch.cuda.synchronize()
    base, rays = ref.image(ref.IMG_MASK).clone(), ref.ray_count()
    # (a) the persistent-wave ray-queue kernel (another mapping of rays to lanes and waves) is an A/B path of the -DHR_DEV_PATHS build:
    #     tests/test_gpu_devpaths.py
    # (b) a BVH over split triangle references: other boxes, duplicated triangles
    monkeypatch.setenv("HR_BVH_SPLIT", "0.02")
    split_scene = hr.Scene(ctx, full["sd"])
    monkeypatch.delenv("HR_BVH_SPLIT")
    assert split_scene.info.tri_bytes > full["scene"].info.tri_bytes
    s = hr.RayTracedShadows(ctx, W, H)
    s.ray_trace(split_scene, _fi(hr, full, 1))
    torch.cuda.synchronize()
    assert torch.equal(s.image(s.IMG_MASK), base) and s.ray_count() == rays
    # (c) reproducibility
    ref.ray_trace(full["scene"], _fi(hr, full, 1))
    torch.cuda.synchronize()
    assert torch.equal(ref.image(ref.IMG_MASK), base)
    lit = int(np.unpackbits(base.cpu().numpy().view(np.uint8)).sum())
    assert 0 < lit <= rays
    for p in (ref, s):
        p.close()
    split_scene.close()


def test_cost_balanced_bands_equal_whole_frame(hr, ctx, full):
    """the multi-GPU decomposition of bench.py at full size: 4 cost-balanced bands, 3 frames with camera motion"""
    import torch
    world = 4
    cal = hr.RayTracedShadows(ctx, W, H)
    cal.ray_trace(full["scene"], _fi(hr, full, 0))
    bounds = tiling.balanced_bounds(tiling.shadow_cost_per_tile_row(full["gbs"][0]["depth"], cal.tile_ray_counts()), world, H)
    cal.close()
    heights = [b - a for a, b in zip(bounds, bounds[1:])]
    assert max(heights) > 1.3 * min(heights)                     # the bands really are unequal
    whole = hr.RayTracedShadows(ctx, W, H)
    bands = [tiling.TiledShadows(ctx, W, H, r, world, bounds=bounds) for r in range(world)]
    for b in bands:
        b.world = 1
    for k in range(3):
        fi = _fi(hr, full, k)
        whole.render(full["scene"], fi)
        for b in bands:
            b.render(full["scene"], fi)
        for r, b in enumerate(bands):
            for peer, (s0, s1), (r0, r1) in tiling.exchange_plan(H, world, r, tiling.HISTORY_HALO, bounds):
                for mine, theirs in zip(b.history_images(k & 1), bands[peer].history_images(k & 1)):
                    mine[r0:r1].copy_(theirs[r0:r1])
        torch.cuda.synchronize()
        ref = whole.output(hr.OUTPUT_ATROUS)
        for r, b in enumerate(bands):
            assert torch.equal(b.pass_.output(hr.OUTPUT_ATROUS)[b.b0:b.b1], ref[b.b0:b.b1]), f"frame {k} band {r}"
    assert sum(int(b.pass_.tile_ray_counts()[b.b0 // 8:(b.b1 + 7) // 8].sum()) for b in bands) == whole.ray_count()


def test_ao_4spp_1080p_matches_oracle(oracle, hr, ctx, full):
    """BASELINE configs[2]: AO at 4 spp, one full-size frame against the oracle (8.2 M rays)"""
    import torch
    osc = oracle.Scene(full["sd"])
    zbp = synth.z_buffer_params()
    op = oracle.AOPass(W, H, spp=4, zbp=zbp)
    gp = hr.RayTracedAO(ctx, W, H, 0)
    gp.params.spp = 4
    host = _host(full["gbs"][0])
    op.render(osc, full["ubos"][0], host, host, full["sob"], full["sr"], 0)
    gp.render(full["scene"], hr.frame_inputs(full["gbs"][0], full["gbs"][0], full["ubos"][0], 0, 0, full["sob_d"], full["sr_d"], z_buffer_params=zbp))
    torch.cuda.synchronize()
    st = op.stages
    mh = (H + 3) // 4
    mask = gp.image(gp.IMG_MASK).cpu().numpy().view(np.uint32)[:4 * mh].reshape(4, mh, -1)
    assert np.array_equal(mask, st["mask"]) and gp.ray_count() == st["rays"] and st["rays"] > 6_000_000
    out = helpers.bits16(gp.output(hr.OUTPUT_UPSAMPLE))
    ref = st["output"] if st["output"].ndim == 2 else st["output"][..., 0]
    assert np.array_equal(out, ref)
    gp.close()
    # the tolerance mode on the same frame (what bench.py's `passes` block times)
    import test_gpu_tolerance as tol
    gf = hr.RayTracedAO(ctx, W, H, 0)
    gf.params.spp = 4
    gf.params.exact = 0
    gf.render(full["scene"], hr.frame_inputs(full["gbs"][0], full["gbs"][0], full["ubos"][0], 0, 0, full["sob_d"], full["sr_d"], z_buffer_params=zbp))
    torch.cuda.synchronize()
    assert np.array_equal(gf.image(gf.IMG_MASK).cpu().numpy().view(np.uint32)[:4 * mh].reshape(4, mh, -1), st["mask"]) and gf.ray_count() == st["rays"]
    ex = tol.tiles_close(gf.image(gf.IMG_TILES).cpu().numpy(), st["tiles"], "AO 1080p (exact = 0)", shape=(H, W))
    tol.compare16(helpers.bits16(gf.output(hr.OUTPUT_UPSAMPLE)), ref, "AO 1080p output (exact = 0)", exclude=ex)
    gf.close()


def test_ddgi_full_grid_shards_equal_unsharded(hr, ctx, full):
    """BASELINE configs[4]: 16x8x16 probes x 256 rays; 2 and 4 z-slab shards (+ emulated all-gather) vs one GPU, 2 frames"""
    import torch
    from hybrid_rendering_amd import api_gi
    lo, hi = full["sd"].bounds()
    u = synth_env.ddgi_uniforms(lo, hi, probe_counts=(16, 8, 16), rays_per_probe=256, normal_bias=0.25)
    sky = synth_env.sky_cubemap(32)
    env = api_gi.environment(torch.from_numpy(sky).cuda().view(torch.float16))
    rng = np.random.RandomState(11)
    orients = [synth_env.random_orientation(rng) for _ in range(2)]
    whole = api_gi.DDGI(ctx, W, H, u)
    for k in range(2):
        whole.render(full["scene"], _fi(hr, full, k), env, orients[k])
    torch.cuda.synchronize()
    wi, wd = (t.clone() for t in whole.current_read())
    wout, wrays = whole.output().clone(), whole.ray_count()
    assert wrays >= 16 * 8 * 16 * 256
    for world in (2, 4):
        gis = [tiling.ShardedDDGI(ctx, W, H, u, r, world) for r in range(world)]
        for k in range(2):
            fi = _fi(hr, full, k)
            for g in gis:
                g.pass_.set_orientation(orients[k])
                g.pass_.ray_trace(full["scene"], fi, env)
                g.pass_.probe_update()
            atl = [g.pass_.current_write() for g in gis]
            for r in range(world):
                for src in range(world):
                    if src != r:
                        for kk, side in ((0, 8), (1, 16)):
                            a, b = tiling.slab_rows(side, *tiling.probe_slabs(16, world, src))
                            atl[r][kk][a:b].copy_(atl[src][kk][a:b])
            for g in gis:
                g.pass_.sample_probe_grid(fi)
                g.pass_.end_frame()
        torch.cuda.synchronize()
        for r, g in enumerate(gis):
            gi, gd = g.pass_.current_read()
            assert torch.equal(gi, wi) and torch.equal(gd, wd), f"world {world} rank {r}: atlases"
            assert torch.equal(g.pass_.output()[g.b0:g.b1], wout[g.b0:g.b1]), f"world {world} rank {r}: sampled band"
        assert sum(g.pass_.ray_count() for g in gis) == wrays
        for g in gis:
            g.pass_.close()
    whole.close()
