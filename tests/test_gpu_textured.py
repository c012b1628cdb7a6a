"""GPU parity with TEXTURED materials (hr_scene_desc.uvs / tangents / material_textures / textures): the closest-hit
shading of DDGI, reflections and the ground-truth path tracer — fetch_albedo / fetch_roughness / fetch_metallic /
fetch_normal of scene_descriptor_set.glsl:133-220 incl. the (T, T, N) normal-map quirk — HIP vs the oracle, every image
bit for bit.  (The oracle's textured path is pinned to the reference's hit shaders by tests/test_ref_shaders.py.)"""
import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cornell", "sponza_small"])
def test_textured_hit_shading(oracle, hr, ctx, name):
    import torch
    from hybrid_rendering_amd import api_gi, api_post, api_reflections
    from oracle import pyoracle_ddgi as od, pyoracle_post as opost, pyoracle_reflections as orf
    W, H = 96, 64
    base = helpers.scene_data(name)
    sd = synth.with_textures(base)
    osc, gsc, gsc_plain = oracle.Scene(sd), hr.Scene(ctx, sd), hr.Scene(ctx, base)
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(4, 3, 4), rays_per_probe=48, normal_bias=1.0 if name == "cornell" else 0.1)
    sky = synth_env.sky_cubemap(8)
    pre, lut = synth_env.prefiltered_chain(sky, 4), synth_env.brdf_lut(8)
    env_np = dict(sky=sky, prefiltered=pre, pre_size=8, pre_levels=4, lut=lut)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(pre), 8, 4, f16(lut))
    frames = helpers.make_frames(oracle, osc, name, W, H, 3, 1.0)
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    g_gi, o_gi = api_gi.DDGI(ctx, W, H, ddgi), od.DDGIPass(ddgi)
    g_rf, o_rf = api_reflections.RayTracedReflections(ctx, W, H, hr.SCALE_FULL_RES), orf.ReflectionsPass(W, H)
    rng = np.random.RandomState(4)
    for f in range(3):
        cur, prev = frames[f]["gb"], frames[f - 1]["gb"] if f else frames[f]["gb"]
        orient = synth_env.random_orientation(rng)
        o_gi.render(osc, frames[f]["ubo"], cur, sky, orient, f)
        irr, dep = o_gi.current_read()
        o_rf.render(osc, frames[f]["ubo"], ddgi, cur, prev, sob, sr, f, env_np, irr, dep, ping_pong=bool(f & 1))
        fi = hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d)
        g_gi.render(gsc, fi, env, orient)
        g_rf.render(gsc, fi, env, g_gi)
        torch.cuda.synchronize()
        a, b = o_gi.stages, o_rf.stages
        assert np.array_equal(helpers.bits16(g_gi.image(g_gi.IMG_RADIANCE)).reshape(a["radiance"].shape), a["radiance"]), f"frame {f}: probe radiance"
        ci, cd = g_gi.current_read()
        assert np.array_equal(helpers.bits16(ci), a["irradiance"]) and np.array_equal(helpers.bits16(cd), a["depth"]), f"frame {f}: atlases"
        assert np.array_equal(helpers.bits16(g_gi.output()), a["output"]), f"frame {f}: sampled irradiance"
        assert np.array_equal(helpers.bits16(g_rf.image(g_rf.IMG_TRACE)), b["trace"]), f"frame {f}: reflection rays"
        assert np.array_equal(helpers.bits16(g_rf.output(hr.OUTPUT_ATROUS)), b["atrous"][-1]), f"frame {f}: reflections a-trous"
    # ground truth: 3 accumulated frames, and the textures must matter
    g_gt, o_gt = api_post.GroundTruthPathTracer(ctx, W, H), opost.GroundTruthPass(W, H)
    for k in range(3):
        o = o_gt.render(osc, frames[0]["ubo"], sky)
        g_gt.render(gsc, frames[0]["ubo"], env)
        torch.cuda.synchronize()
        assert np.array_equal(helpers.bits16(g_gt.output()), o), f"ground truth frame {k}"
    g_plain = api_post.GroundTruthPathTracer(ctx, W, H)
    g_first = api_post.GroundTruthPathTracer(ctx, W, H)
    g_plain.render(gsc_plain, frames[0]["ubo"], env)
    g_first.render(gsc, frames[0]["ubo"], env)
    torch.cuda.synchronize()
    assert (helpers.bits16(g_plain.output()) != helpers.bits16(g_first.output())).any(-1).mean() > 0.1
    for p in (g_gi, g_rf, g_gt, g_plain, g_first, gsc, gsc_plain):
        p.close()
