"""Instanced scenes on the GPU (hr_scene_create_instanced / hr_scene_update_instances: scene_descriptor_set.glsl:30-34,102-160; main.cpp:74
build_tlas every frame): one world-space 8-wide BVH with a private subtree per instance, vertices transformed and every node box refitted on
the GPU each update.  Against the oracle's instanced scene (bit-exact masks, hit records, trace images — the oracle's instanced hit shading is
pinned to the reference's own hit shaders by tests/test_instances.py) and against a flattened hr_scene_create over the same world vertices."""
import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env

pytestmark = pytest.mark.gpu


def _rays(n, seed, lo=5.0, hi=95.0):
    rng = np.random.RandomState(seed)
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    r[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    r[:, 3], r[:, 7] = 1e4, 0.01
    r[: n // 4, 3] = rng.uniform(3, 40, n // 4)     # short rays as well
    return r


def _mats(isd, n_boxes, seed, frame):
    return synth.InstancedSceneData(isd.meshes, synth.instanced_cornell_instances(n_boxes, seed=seed, frame=frame), isd.materials).matrices()


@pytest.mark.parametrize("n_boxes,seed", [(5, 3), (70, 9), (600, 4)])
def test_queries_after_every_update_equal_the_flattened_scene_and_the_oracle(oracle, hr, ctx, n_boxes, seed):
    """6 / 71 / 601 instances (one, three and four top levels), half of them moving every frame: any-hit and closest-hit records (t, u, v, triangle)
    of 40 k rays, bit for bit, against hr_scene_create over the flattened vertices and against the oracle"""
    import torch
    isd = synth.instanced_cornell(n_boxes, seed=seed)
    g = hr.InstancedScene(ctx, isd)
    rays = _rays(40000, seed)
    rd = torch.from_numpy(rays).cuda()
    for f in (0, 1, 2, 7, 90, 91):   # frame 90: the movers have crossed the room — the top level is re-built on its own when that pays
        mats = _mats(isd, n_boxes, seed, f)
        g.update(mats)
        if f == 91:
            g.rebuild_top_level()   # and on demand: the instance roots change slots, every level is refitted
        flat_sd = isd.flatten(mats)
        gf = hr.Scene(ctx, flat_sd)
        occ, (tuv, prim) = g.any_hit(rd).cpu().numpy(), [t.cpu().numpy() for t in g.closest_hit(rd)]
        occ_f, (tuv_f, prim_f) = gf.any_hit(rd).cpu().numpy(), [t.cpu().numpy() for t in gf.closest_hit(rd)]
        assert np.array_equal(occ, occ_f), f"frame {f}: any-hit differs from the flattened scene on {int((occ != occ_f).sum())} rays"
        assert np.array_equal(prim, prim_f) and np.array_equal(tuv.view(np.uint32), tuv_f.view(np.uint32)), f"frame {f}: closest hits differ from the flattened scene"
        if n_boxes <= 70:
            osc = oracle.Scene(flat_sd)
            assert np.array_equal(occ, osc.any_hit(rays))
            tuv_o, prim_o = osc.closest_hit(rays)
            assert np.array_equal(prim, prim_o) and np.array_equal(tuv.view(np.uint32)[prim >= 0], tuv_o.view(np.uint32)[prim_o >= 0])
        info, finfo = g.refresh_info(), gf.info
        assert list(info.bounds_lo) == list(finfo.bounds_lo) and list(info.bounds_hi) == list(finfo.bounds_hi), "exact bounds after the update"
        gf.close()
        assert 0.05 < occ.mean() < 0.999
    assert g.top_level_rebuilds >= (1 if n_boxes > 1 else 0)
    g.close()


def test_passes_on_a_scene_whose_instances_move_every_frame(oracle, hr, ctx):
    """shadows (masks + every denoise image, exact mode), AO 2 spp masks, DDGI ray images + atlases, reflections trace image and ground truth, 4
    frames, every second instance moving each frame: bit-identical to the oracle's instanced scene; the G-buffer synthesiser too"""
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections, api_post
    from oracle import pyoracle_ddgi as od, pyoracle_reflections as orf
    n_boxes, seed, W, H = 9, 5, 160, 120
    isd = synth.instanced_cornell(n_boxes, seed=seed)
    g, osc = hr.InstancedScene(ctx, isd), oracle.InstancedScene(isd)
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    lo, hi = isd.flatten().bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(4, 3, 4), rays_per_probe=64, normal_bias=1.0)
    sky = synth_env.sky_cubemap(16)
    pre, lut = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(16)
    env_np = dict(sky=sky, prefiltered=pre, pre_size=16, pre_levels=5, lut=lut)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(pre), 16, 5, f16(lut))
    gs, os_ = hr.RayTracedShadows(ctx, W, H), oracle.ShadowsPass(W, H)
    ga, oa = hr.RayTracedAO(ctx, W, H, 0), oracle.AOPass(W, H, spp=2, zbp=synth.z_buffer_params())
    ga.params.spp = 2
    gd, odd = api_gi.DDGI(ctx, W, H, ddgi), od.DDGIPass(ddgi)
    gr, orr = api_reflections.RayTracedReflections(ctx, W, H, 0), orf.ReflectionsPass(W, H)
    cams = helpers.cameras("cornell", W / H, 5, 1.0)
    light = helpers.light_for("cornell", "soft")
    rng = np.random.RandomState(2)
    prev_np = None
    for f in range(4):
        mats = _mats(isd, n_boxes, seed, f)
        g.update(mats)
        osc.update(mats)
        ubo = synth.make_ubo(cams[f], cams[f - 1] if f else None, light)
        cur_d = g.gbuffer(ubo, W, H)
        cur = osc.gbuffer(ubo, W, H)
        for k in cur:
            got = cur_d[k].cpu().numpy()
            assert np.array_equal(got.view(np.uint16) if got.dtype == np.float16 else got, cur[k]), f"frame {f}: G-buffer {k} of the instanced scene"
        ch = cur["gb3"][..., 0]
        ch[ch == np.float16(0.8).view(np.uint16)] = np.float16(0.03).view(np.uint16)   # mirrors everywhere: every instance is reflected
        cur_d = helpers.to_cuda(cur)
        prev = prev_np if prev_np is not None else cur
        fi = hr.frame_inputs(cur_d, helpers.to_cuda(prev), ubo, f, f & 1, sob_d, sr_d, z_buffer_params=synth.z_buffer_params())
        gs.render(g, fi)
        os_.render(osc, ubo, cur, prev, sob, sr, f)
        torch.cuda.synchronize()
        assert np.array_equal(gs.image(gs.IMG_MASK).cpu().numpy().view(np.uint32), os_.stages["mask"]), f"frame {f}: shadow mask"
        assert np.array_equal(helpers.bits16(gs.output(hr.OUTPUT_ATROUS)), os_.stages["output"]), f"frame {f}: denoised shadows"
        ga.render(g, fi)
        oa.render(osc, ubo, cur, prev, sob, sr, f)
        torch.cuda.synchronize()
        mh = (H + 3) // 4
        assert np.array_equal(ga.image(ga.IMG_MASK).cpu().numpy().view(np.uint32)[:2 * mh].reshape(2, mh, -1), oa.stages["mask"]), f"frame {f}: AO masks (entry-node table rebuilt after the update)"
        orient = synth_env.random_orientation(rng)
        gd.render(g, fi, env, orient)
        odd.render(osc, ubo, cur, sky, orient, f)
        torch.cuda.synchronize()
        assert np.array_equal(helpers.bits16(gd.image(gd.IMG_RADIANCE)), odd.stages["radiance"]), f"frame {f}: DDGI radiance (instanced hit shading)"
        assert np.array_equal(helpers.bits16(gd.image(gd.IMG_DIRDIST)), odd.stages["direction_distance"])
        irr, dep = odd.current_read()
        gi, gdp = gd.current_read()
        assert np.array_equal(helpers.bits16(gi), irr) and np.array_equal(helpers.bits16(gdp), dep)
        gr.render(g, fi, env, gd)
        orr.render(osc, ubo, ddgi, cur, prev, sob, sr, f, env_np, irr, dep, ping_pong=bool(f & 1))
        torch.cuda.synchronize()
        assert np.array_equal(helpers.bits16(gr.image(gr.IMG_TRACE)), orr.stages["trace"]), f"frame {f}: reflections trace image (transform_vertex at the hit)"
        assert gr.ray_count() == orr.stages["rays"]
        assert np.array_equal(helpers.bits16(gr.output(hr.OUTPUT_ATROUS)), orr.stages["atrous"][-1])
        prev_np = cur
    for p in (gs, ga, gd, gr, g):
        p.close()


def test_one_identity_instance_is_the_flat_scene(oracle, hr, ctx):
    """hr_scene_create_instanced with ONE identity instance against hr_scene_create of the same triangles: masks, denoised images and the reflections
    trace image are the same bits (nothing moves: the flattened path and the instanced path agree)"""
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections
    sd = helpers.scene_data("sponza_small")
    one = synth.InstancedSceneData(meshes=[sd], instances=[(synth.model_matrix(), 0, 1)], materials=sd.materials)
    gi, gf = hr.InstancedScene(ctx, one), hr.Scene(ctx, sd)
    assert gi.info.n_nodes == gf.info.n_nodes and gi.info.n_tris == gf.info.n_tris and list(gi.info.bounds_lo) == list(gf.info.bounds_lo)
    W, H = 256, 144
    osc = oracle.Scene(sd)
    frames = helpers.make_frames(oracle, osc, "sponza_small", W, H, 3, 1.0)
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
    sky = synth_env.sky_cubemap(16)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 16, 5, f16(synth_env.brdf_lut(16)))
    passes = {}
    for tag, sc in (("inst", gi), ("flat", gf)):
        passes[tag] = (sc, hr.RayTracedShadows(ctx, W, H), hr.RayTracedAO(ctx, W, H, 0), api_gi.DDGI(ctx, W, H, ddgi), api_reflections.RayTracedReflections(ctx, W, H, 0))
    rng = np.random.RandomState(5)
    for f in range(3):
        cur, prev = frames[f]["gb"], frames[f - 1 if f else 0]["gb"]
        fi = hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d, z_buffer_params=synth.z_buffer_params())
        orient = synth_env.random_orientation(rng)
        out = {}
        for tag, (sc, ps, pa, pd, pr) in passes.items():
            ps.render(sc, fi); pa.render(sc, fi); pd.render(sc, fi, env, orient); pr.render(sc, fi, env, pd)
            torch.cuda.synchronize()
            out[tag] = [ps.image(ps.IMG_MASK).cpu().numpy(), helpers.bits16(ps.output(hr.OUTPUT_ATROUS)), pa.image(pa.IMG_MASK).cpu().numpy(), helpers.bits16(pd.image(pd.IMG_RADIANCE)),
                        helpers.bits16(pr.image(pr.IMG_TRACE)), helpers.bits16(pr.output(hr.OUTPUT_ATROUS))]
        for a, b, what in zip(out["inst"], out["flat"], ("shadow mask", "denoised shadows", "AO masks", "DDGI radiance", "reflections trace", "denoised reflections")):
            assert np.array_equal(a, b), f"frame {f}: {what}"
    for sc, *ps in passes.values():
        for p in ps:
            p.close()
        sc.close()


def test_update_rejects_flat_scenes_and_bad_matrices(hr, ctx):
    import ctypes as C
    sd = synth.cornell32()
    flat = hr.Scene(ctx, sd)
    m = np.eye(4, dtype=np.float32).reshape(16)
    assert hr.lib().hr_scene_update_instances(flat.h, m.ctypes.data_as(C.POINTER(C.c_float)), None) == 1   # HR_ERR_INVALID_ARG
    assert hr.lib().hr_scene_instance_count(flat.h) == 0
    isd = synth.instanced_cornell(3)
    g = hr.InstancedScene(ctx, isd)
    assert hr.lib().hr_scene_instance_count(g.h) == 4 and g.id != flat.id
    bad = isd.matrices().copy()
    bad[2, 5] = np.nan
    assert hr.lib().hr_scene_update_instances(g.h, bad.ctypes.data_as(C.POINTER(C.c_float)), None) == 1
    g.update(isd.matrices())      # still usable
    flat.close(); g.close()


def test_degenerate_instances_against_the_flattened_scene(hr, ctx):
    """an empty mesh, an instance squashed to a plane (scale 0 along one axis), one collapsed to a point, one far away, one mirrored (negative
    scale), mixed with ordinary ones: queries equal those of hr_scene_create over the same world vertices, before and after an update"""
    import dataclasses
    import torch
    base = synth.instanced_cornell(4, seed=8)
    empty = dataclasses.replace(base.meshes[2], verts=np.zeros((0, 3, 3), np.float32), normals=np.zeros((0, 3, 3), np.float32), tri_material=np.zeros(0, np.uint32),
                                tri_mesh_id=np.zeros(0, np.uint32))
    meshes = list(base.meshes) + [empty]
    inst = list(base.instances)
    inst += [(synth.model_matrix((50, 50, 50)), 3, 30),                                   # the empty mesh
             (synth.model_matrix((30, 20, 60), (0, 1, 0), 0.4, (20, 0.0, 20)), 1, 31),    # squashed to a plane
             (synth.model_matrix((70, 20, 30), (1, 0, 0), 0.0, 0.0), 2, 32),              # a point
             (synth.model_matrix((4000, 3000, -2500), (1, 1, 0), 1.1, 15.0), 1, 33),      # far away
             (synth.model_matrix((60, 30, 40), (0, 0, 1), 0.7, (-12, 9, 14)), 2, 34)]     # mirrored
    isd = synth.InstancedSceneData(meshes=meshes, instances=inst, materials=base.materials)
    g = hr.InstancedScene(ctx, isd)
    rays = _rays(30000, 12)
    rays[:3000, :3] = np.array([3900, 2950, -2450], np.float32) + np.random.RandomState(1).uniform(-60, 60, (3000, 3)).astype(np.float32)   # around the far one
    rd = torch.from_numpy(rays).cuda()
    mats = isd.matrices()
    for step in range(3):
        if step:
            mats = mats.copy()
            mats[1:5, 12:15] += np.float32(3.5 * step)      # move the ordinary ones
            mats[8, 12:15] += np.float32(-500.0 * step)     # and the far one
            g.update(mats)
        gf = hr.Scene(ctx, isd.flatten(mats))
        occ, (tuv, prim) = g.any_hit(rd).cpu().numpy(), [t.cpu().numpy() for t in g.closest_hit(rd)]
        occ_f, (tuv_f, prim_f) = gf.any_hit(rd).cpu().numpy(), [t.cpu().numpy() for t in gf.closest_hit(rd)]
        assert np.array_equal(occ, occ_f) and np.array_equal(prim, prim_f) and np.array_equal(tuv.view(np.uint32), tuv_f.view(np.uint32)), f"step {step}"
        assert (prim >= 0).mean() > 0.5
        gf.close()
    g.close()


def test_textured_instances(oracle, hr, ctx):
    """textured materials on instanced meshes: per-MESH texture coordinates and tangents (object space), the normal map's TBN built from the tangent
    that transform_vertex rotated (scene_descriptor_set.glsl:150-160, :168-220) — DDGI radiance, reflections trace image and the ground-truth
    accumulator against the oracle's instanced scene (which tests/test_instances.py pins to the reference's own hit shaders), two moving frames"""
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections, api_post
    from oracle import pyoracle_ddgi as od, pyoracle_reflections as orf, pyoracle_post as opost
    n_boxes, seed, W, H = 7, 6, 128, 96
    isd = synth.instanced_cornell(n_boxes, seed=seed, textured=True)
    g, osc = hr.InstancedScene(ctx, isd), oracle.InstancedScene(isd)
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    lo, hi = isd.flatten().bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(3, 3, 3), rays_per_probe=64, normal_bias=1.0)
    sky = synth_env.sky_cubemap(16)
    pre, lut = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(16)
    env_np = dict(sky=sky, prefiltered=pre, pre_size=16, pre_levels=5, lut=lut)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(pre), 16, 5, f16(lut))
    gd, odd = api_gi.DDGI(ctx, W, H, ddgi), od.DDGIPass(ddgi)
    gr, orr = api_reflections.RayTracedReflections(ctx, W, H, 0), orf.ReflectionsPass(W, H)
    gt, ogt = api_post.GroundTruthPathTracer(ctx, W, H), opost.GroundTruthPass(W, H)
    cams = helpers.cameras("cornell", W / H, 3, 1.0)
    light = helpers.light_for("cornell", "soft")
    rng = np.random.RandomState(4)
    prev = None
    for f in range(2):
        mats = synth.InstancedSceneData(isd.meshes, synth.instanced_cornell_instances(n_boxes, seed=seed, frame=f), isd.materials).matrices()
        g.update(mats)
        osc.update(mats)
        ubo = synth.make_ubo(cams[f], cams[f - 1] if f else None, light)
        cur = osc.gbuffer(ubo, W, H)
        ch = cur["gb3"][..., 0]
        ch[ch == np.float16(0.8).view(np.uint16)] = np.float16(0.03).view(np.uint16)
        cur_d = helpers.to_cuda(cur)
        fi = hr.frame_inputs(cur_d, helpers.to_cuda(prev if prev is not None else cur), ubo, f, f & 1, sob_d, sr_d)
        orient = synth_env.random_orientation(rng)
        gd.render(g, fi, env, orient)
        odd.render(osc, ubo, cur, sky, orient, f)
        torch.cuda.synchronize()
        assert np.array_equal(helpers.bits16(gd.image(gd.IMG_RADIANCE)), odd.stages["radiance"]), f"frame {f}: DDGI radiance"
        irr, dep = odd.current_read()
        gr.render(g, fi, env, gd)
        orr.render(osc, ubo, ddgi, cur, prev if prev is not None else cur, sob, sr, f, env_np, irr, dep, ping_pong=bool(f & 1))
        torch.cuda.synchronize()
        assert np.array_equal(helpers.bits16(gr.image(gr.IMG_TRACE)), orr.stages["trace"]), f"frame {f}: reflections trace image"
        gt.render(g, ubo, env)
        ref = ogt.render(osc, ubo, sky)
        torch.cuda.synchronize()
        assert np.array_equal(helpers.bits16(gt.output()), ref), f"frame {f}: ground-truth accumulator"
        prev = cur
    for p in (gd, gr, gt, g):
        p.close()
