"""GPU vs the COMMITTED golden fixtures (not the live oracle): the HIP passes reproduce tests/golden/*.npz bit for bit."""
import os

import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gold(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def test_shadows_and_ao_golden(oracle, hr, ctx):
    import torch
    sd = helpers.scene_data("cornell")
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    w = h = 64
    frames = helpers.make_frames(oracle, osc, "cornell", w, h, 3, 1.5, "soft")   # inputs only
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    sh, ao = hr.RayTracedShadows(ctx, w, h), hr.RayTracedAO(ctx, w, h, hr.SCALE_HALF_RES)
    zbp = synth.z_buffer_params()
    for f in range(3):
        cur, prev = frames[f]["gb"], frames[f - 1]["gb"] if f else frames[f]["gb"]
        cur_d, prev_d = helpers.to_cuda(cur), helpers.to_cuda(prev)
        sh.render(gsc, hr.frame_inputs(cur_d, prev_d, frames[f]["ubo"], f, f & 1, sob_d, sr_d))
        ao.render(gsc, hr.frame_inputs(helpers.to_cuda(helpers.nearest_mip(cur, 1)), helpers.to_cuda(helpers.nearest_mip(prev, 1)), frames[f]["ubo"], f, f & 1,
                                       sob_d, sr_d, cur_full=cur_d, z_buffer_params=zbp))
    torch.cuda.synchronize()
    g = _gold("shadows_cornell64")
    assert np.array_equal(frames[2]["gb"]["gb2"], g["gb2"]) and np.array_equal(frames[2]["gb"]["depth"], g["depth"])
    assert np.array_equal(sh.image(sh.IMG_MASK).cpu().numpy().view(np.uint32), g["mask"])
    assert np.array_equal(helpers.bits16(sh.image(sh.IMG_TEMPORAL)), g["temporal"])
    assert np.array_equal(helpers.bits16(sh.image(sh.IMG_MOMENTS0)), g["moments"])          # frame 2: ping_pong = 0
    assert np.array_equal(sh.image(sh.IMG_TILES).cpu().numpy(), g["tiles"])
    assert np.array_equal(helpers.bits16(sh.output(hr.OUTPUT_ATROUS)), g["output"])
    a = _gold("ao_cornell64_half")
    assert np.array_equal(ao.image(ao.IMG_MASK).cpu().numpy().view(np.uint32)[:a["mask"].shape[1]], a["mask"][0])
    assert np.array_equal(helpers.bits16(ao.image(ao.IMG_AO0)), a["temporal"])
    assert np.array_equal(helpers.bits16(ao.image(ao.IMG_BLUR1)), a["blur1"])
    assert np.array_equal(helpers.bits16(ao.output(hr.OUTPUT_UPSAMPLE)), a["output"][..., 0])


def test_ddgi_and_reflections_golden(oracle, hr, ctx):
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections
    sd = helpers.scene_data("sponza_small")
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    w, h = 48, 32
    frames = helpers.make_frames(oracle, osc, "sponza_small", w, h, 2, 1.0)
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(3, 2, 3), rays_per_probe=32, normal_bias=0.1)
    sky = synth_env.sky_cubemap(8)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 4)), 8, 4, f16(synth_env.brdf_lut(8)))
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    gd, gr = api_gi.DDGI(ctx, w, h, ddgi), api_reflections.RayTracedReflections(ctx, w, h, hr.SCALE_FULL_RES)
    rng = np.random.RandomState(3)
    for f in range(2):
        cur_d = helpers.to_cuda(frames[f]["gb"])
        prev_d = helpers.to_cuda(frames[f - 1]["gb"] if f else frames[f]["gb"])
        fi = hr.frame_inputs(cur_d, prev_d, frames[f]["ubo"], f, f & 1, sob_d, sr_d)
        gd.render(gsc, fi, env, synth_env.random_orientation(rng))
        gr.set_camera_delta((-1.0, 0, 0) if f else (0, 0, 0))
        gr.render(gsc, fi, env, gd)
    torch.cuda.synchronize()
    d, r = _gold("ddgi_sponza"), _gold("reflections_sponza")
    assert np.array_equal(helpers.bits16(gd.image(gd.IMG_RADIANCE)).reshape(d["radiance"].shape), d["radiance"])
    assert np.array_equal(helpers.bits16(gd.image(gd.IMG_DIRDIST)).reshape(d["direction_distance"].shape), d["direction_distance"])
    ci, cd = gd.current_read()
    assert np.array_equal(helpers.bits16(ci), d["irradiance"]) and np.array_equal(helpers.bits16(cd), d["depth"])
    assert np.array_equal(helpers.bits16(gd.output()), d["output"])
    assert np.array_equal(helpers.bits16(gr.image(gr.IMG_TRACE)), r["trace"])
    assert np.array_equal(helpers.bits16(gr.image(gr.IMG_COLOR1)), r["temporal"])                # frame 1: ping_pong = 1
    assert np.array_equal(gr.image(gr.IMG_TILES).cpu().numpy(), r["tiles"])
    assert np.array_equal(helpers.bits16(gr.output(hr.OUTPUT_UPSAMPLE)), r["output"])
