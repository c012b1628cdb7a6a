"""End-to-end hybrid frame: shadows + AO + DDGI + reflections + the deferred composite, HIP vs oracle, final HDR image
bit for bit over several frames ('lets end-to-end image diffs replace per-buffer diffs')."""
import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags", [0b1111, 0b0011, 0b0000])
def test_hybrid_frame_end_to_end(oracle, hr, ctx, flags):
    import torch
    from hybrid_rendering_amd import api_deferred, api_gi, api_reflections
    from oracle import pyoracle_ddgi as od, pyoracle_deferred as odf, pyoracle_reflections as orf
    name, W, H, n_frames = "sponza_small", 192, 112, 3
    sd = helpers.scene_data(name)
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
    sky = synth_env.sky_cubemap(16)
    pre, lut, sh9 = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(16), synth_env.sh9_from_cubemap(sky)
    env_np = dict(sky=sky, prefiltered=pre, pre_size=16, pre_levels=5, lut=lut)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(pre), 16, 5, f16(lut))
    frames = helpers.make_frames(oracle, osc, name, W, H, n_frames, 1.5, scale_mips=1)
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    zbp = synth.z_buffer_params()
    # reference defaults: shadows full res, AO + reflections half res (ray_traced_*.h), DDGI sample full res
    g_sh, g_ao = hr.RayTracedShadows(ctx, W, H), hr.RayTracedAO(ctx, W, H, hr.SCALE_HALF_RES)
    g_gi, g_rf = api_gi.DDGI(ctx, W, H, ddgi), api_reflections.RayTracedReflections(ctx, W, H, hr.SCALE_HALF_RES)
    g_df = api_deferred.DeferredShading(ctx, W, H)
    g_df.set_sh9(sh9)
    g_df.params.use_ray_traced_shadows, g_df.params.use_ray_traced_ao = flags & 1, (flags >> 1) & 1
    g_df.params.use_ray_traced_reflections, g_df.params.use_ddgi = (flags >> 2) & 1, (flags >> 3) & 1
    o_sh, o_ao = oracle.ShadowsPass(W, H), oracle.AOPass(W // 2, H // 2, zbp=zbp)
    o_gi, o_rf = od.DDGIPass(ddgi), orf.ReflectionsPass(W // 2, H // 2)
    rng = np.random.RandomState(11)
    ping = False
    for f in range(n_frames):
        full, pfull = frames[f]["gb"], frames[f - 1]["gb"] if f else frames[f]["gb"]
        half, phalf = frames[f]["mips"][1], (frames[f - 1] if f else frames[f])["mips"][1]
        ubo, orient = frames[f]["ubo"], synth_env.random_orientation(rng)
        # ---- oracle frame (main.cpp:79-88 order)
        sh = o_sh.render(osc, ubo, full, pfull, sob, sr, f)
        ao = o_ao.render(osc, ubo, half, phalf, sob, sr, f, full=full)
        gi = o_gi.render(osc, ubo, full, sky, orient, f)
        irr, dep = o_gi.current_read()
        rf = o_rf.render(osc, ubo, ddgi, half, phalf, sob, sr, f, env_np, irr, dep, camera_delta=(-1.5, 0, 0) if f else (0, 0, 0), full=full, ping_pong=ping)
        ref = odf.shade(ubo, full, sh, ao, rf, gi, flags, sh9, env_np)
        # ---- GPU frame
        full_d, pfull_d, half_d, phalf_d = (helpers.to_cuda(g) for g in (full, pfull, half, phalf))
        fi_full = hr.frame_inputs(full_d, pfull_d, ubo, f, ping, sob_d, sr_d, z_buffer_params=zbp)
        fi_half = hr.frame_inputs(half_d, phalf_d, ubo, f, ping, sob_d, sr_d, cur_full=full_d, z_buffer_params=zbp)
        g_sh.render(gsc, fi_full)
        g_ao.render(gsc, fi_half)
        g_gi.render(gsc, fi_full, env, orient)
        g_rf.set_camera_delta((-1.5, 0, 0) if f else (0, 0, 0))
        g_rf.render(gsc, fi_half, env, g_gi)
        g_df.render(fi_full, env, shadow=g_sh.output(hr.OUTPUT_UPSAMPLE), ao=g_ao.output(hr.OUTPUT_UPSAMPLE),
                    reflections=g_rf.output(hr.OUTPUT_UPSAMPLE), gi=g_gi.output())
        torch.cuda.synchronize()
        got = helpers.bits16(g_df.output())
        assert np.array_equal(got, ref), f"flags {flags:04b} frame {f}: final image differs in {(got != ref).sum()} halfs"
        ping = not ping
    img = oracle.f16(ref[..., :3])
    assert np.isfinite(img).all() and img.mean() > 0.01


def test_skybox_covers_exactly_the_sky_texels(oracle, hr, ctx):
    """DeferredShading::render = render_shading + render_skybox: texels at depth 1 take the sky cubemap along the ray through
    the pixel centre (draw_skybox = 1, the default); with draw_skybox = 0 they keep the (meaningless) shaded value"""
    import torch
    from hybrid_rendering_amd import api_deferred, api_gi
    from oracle import pyoracle_deferred as odf
    name, W, H = "sponza_small", 160, 96
    sd = helpers.scene_data(name)
    osc = oracle.Scene(sd)
    fr = helpers.make_frames(oracle, osc, name, W, H, 1, 0.0)[0]
    sky = synth_env.sky_cubemap(16)
    pre, lut, sh9 = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(16), synth_env.sh9_from_cubemap(sky)
    env_np = dict(sky=sky, prefiltered=pre, pre_size=16, pre_levels=5, lut=lut)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(pre), 16, 5, f16(lut))
    sob, sr = synth.blue_noise_tables()
    fi = hr.frame_inputs(helpers.to_cuda(fr["gb"]), None, fr["ubo"], 0, 0, torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda())
    g = api_deferred.DeferredShading(ctx, W, H)
    g.set_sh9(sh9)
    g.params.use_ray_traced_shadows = g.params.use_ray_traced_ao = g.params.use_ray_traced_reflections = g.params.use_ddgi = 0
    sky_px = fr["gb"]["depth"] == 1.0
    assert 0.005 < sky_px.mean() < 0.9
    outs = {}
    for draw in (1, 0):
        g.params.draw_skybox = draw
        g.render(fi, env)
        torch.cuda.synchronize()
        outs[draw] = helpers.bits16(g.output()).copy()
        assert np.array_equal(outs[draw], odf.shade(fr["ubo"], fr["gb"], None, None, None, None, 0, sh9, env_np, skybox=bool(draw))), f"draw_skybox {draw}"
    assert np.array_equal(outs[1][~sky_px], outs[0][~sky_px]) and (outs[1][sky_px] != outs[0][sky_px]).any()
    col = oracle.f16(outs[1][sky_px][:, :3])
    assert col[:, 2].mean() > col[:, 0].mean()          # the procedural sky is blue
    g.close()
