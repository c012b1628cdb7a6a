"""GPU parity for the SURVEY §8f rows 3-4: GroundTruthPathTracer and TemporalAA (HIP, through the C ABI) vs the CPU
oracle and the committed golden fixtures, bit for bit."""
import os
import sys

import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _env(sky):
    import torch
    from hybrid_rendering_amd import api_gi
    return api_gi.environment(torch.from_numpy(sky).cuda().view(torch.float16))


@pytest.mark.parametrize("name,w,h,light,rm", [("cornell", 96, 80, "soft", 1.0), ("sponza_small", 160, 96, "default", 0.5), ("sponza_small", 72, 40, "point", 1.0)])
def test_ground_truth_matches_oracle(oracle, hr, ctx, name, w, h, light, rm):
    import torch
    from hybrid_rendering_amd import api_post
    from oracle import pyoracle_post as op
    sd = helpers.scene_data(name)
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    frames = helpers.make_frames(oracle, osc, name, w, h, 2, 1.0, light)
    sky = synth_env.sky_cubemap(16)
    env = _env(sky)
    gp, o = api_post.GroundTruthPathTracer(ctx, w, h), op.GroundTruthPass(w, h, roughness_multiplier=rm)
    gp.params.roughness_multiplier = rm
    for f in range(5):
        ubo = frames[0]["ubo"] if f < 3 else frames[1]["ubo"]
        if f == 3:                       # camera moved: the application restarts the accumulation
            gp.restart_accumulation(); o.restart_accumulation()
        ref = o.render(osc, ubo, sky)
        gp.render(gsc, ubo, env)
        torch.cuda.synchronize()
        got = helpers.bits16(gp.output())
        assert np.array_equal(got, ref), f"frame {f}: {(got != ref).sum()} halfs differ"
        assert gp.ray_count() == o.rays
    img = ref.view(np.float16).astype(np.float32)[..., :3]
    assert img.mean() > 0.02 and img.std() > 0.01
    gp.close(); gsc.close()


def test_ground_truth_bands_and_golden(oracle, hr, ctx):
    import torch
    import make_golden
    from hybrid_rendering_amd import api_post
    sd = helpers.scene_data("sponza_small")
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    w, h = 48, 32
    frames = helpers.make_frames(oracle, osc, "sponza_small", w, h, 2, 1.0)
    sky = synth_env.sky_cubemap(8)
    env = _env(sky)
    whole = api_post.GroundTruthPathTracer(ctx, w, h)
    bands = [api_post.GroundTruthPathTracer(ctx, w, h, band=(0, 16, 0, 0)), api_post.GroundTruthPathTracer(ctx, w, h, band=(16, 32, 0, 0))]
    for f in range(3):
        for p in [whole] + bands:
            p.render(gsc, frames[0]["ubo"], env)
    torch.cuda.synchronize()
    g = np.load(os.path.join(HERE, "golden", "ground_truth_sponza.npz"))
    out = helpers.bits16(whole.output())
    assert np.array_equal(out, g["output"]) and whole.ray_count() == int(g["rays"][0])
    assert np.array_equal(helpers.bits16(bands[0].output())[:16], out[:16]) and np.array_equal(helpers.bits16(bands[1].output())[16:], out[16:])
    assert bands[0].ray_count() + bands[1].ray_count() == whole.ray_count()


@pytest.mark.parametrize("reset,sharpen", [(True, True), (False, True), (False, False)])
def test_taa_matches_oracle(oracle, hr, ctx, reset, sharpen):
    import torch
    import make_golden
    from hybrid_rendering_amd import api_post
    from oracle import pyoracle_post as op
    name, w, h, n = "sponza_small", 176, 104, 4
    sd = helpers.scene_data(name)
    osc = oracle.Scene(sd)
    frames = helpers.make_frames(oracle, osc, name, w, h, n, 2.0)
    gp, o = api_post.TemporalAA(ctx, w, h), op.TAAPass(w, h, reset=reset, sharpen=sharpen)
    gp.params.reset, gp.params.sharpen = int(reset), int(sharpen)
    ping = False
    for f in range(n):
        color = make_golden.hdr_color(frames[f]["gb"])
        j_ref, j = o.update(f), gp.update(f)
        assert np.array_equal(j_ref.view(np.uint32), j.view(np.uint32)), f"frame {f}: jitter {j} vs {j_ref}"
        o.render(color, frames[f]["gb"], ping)
        gp.render(torch.from_numpy(color).cuda().view(torch.float16), helpers.to_cuda(frames[f]["gb"]), ping)
        torch.cuda.synchronize()
        got, ref = helpers.bits16(gp.output(ping)), o.output(ping)
        assert np.array_equal(got, ref), f"frame {f}: {(got != ref).sum()} halfs differ"
        ping = not ping
    mv = frames[-1]["gb"]["gb2"][..., 2:].view(np.float16).astype(np.float32)
    assert np.abs(mv).max() > 0          # the camera moves: the history tap is displaced
    gp.close()


def test_taa_golden_and_disabled(oracle, hr, ctx):
    import torch
    import make_golden
    from hybrid_rendering_amd import api_post
    sd = helpers.scene_data("sponza_small")
    osc = oracle.Scene(sd)
    w, h = 48, 32
    frames = helpers.make_frames(oracle, osc, "sponza_small", w, h, 2, 1.0)
    gp = api_post.TemporalAA(ctx, w, h)
    gp.params.reset = 0
    for f in range(2):
        j = gp.update(f)
        gp.render(torch.from_numpy(make_golden.hdr_color(frames[f]["gb"])).cuda().view(torch.float16), helpers.to_cuda(frames[f]["gb"]), f & 1)
    torch.cuda.synchronize()
    g = np.load(os.path.join(HERE, "golden", "taa_sponza.npz"))
    assert np.array_equal(j.view(np.uint32), g["jitter"].view(np.uint32))
    assert np.array_equal(helpers.bits16(gp.output(1)), g["output"])
    before = helpers.bits16(gp.output(0)).copy()
    gp.params.enabled = 0                  # temporal_aa.cpp:94: render() is a no-op, update() zeroes the jitter
    assert np.all(gp.update(7) == 0)
    gp.render(torch.from_numpy(make_golden.hdr_color(frames[0]["gb"])).cuda().view(torch.float16), helpers.to_cuda(frames[0]["gb"]), 0)
    torch.cuda.synchronize()
    assert np.array_equal(helpers.bits16(gp.output(0)), before)


def test_tone_map_matches_oracle(oracle, hr, ctx):
    """hr_tone_map (ToneMap::render): fp32 FS_OUT_Color bit for bit, UNORM8 = floor(c * 255 + 0.5)"""
    import torch
    from hybrid_rendering_amd import api_post
    from oracle import pyoracle_post as opost
    rng = np.random.RandomState(2)
    W, H = 203, 117
    col = rng.uniform(0.0, 6.0, (H, W, 4)).astype(np.float16)
    col[0, :8, :3] = [0.0, 1e-7, 60000.0]
    col = np.ascontiguousarray(col)
    col_d = torch.from_numpy(col).cuda()
    for single, exposure in ((False, 1.0), (False, 0.37), (True, 1.0)):
        ref = opost.tone_map(col.view(np.uint16), single, exposure)
        f, b = api_post.tone_map(ctx, col_d, single, exposure)
        torch.cuda.synchronize()
        assert np.array_equal(f.cpu().numpy().view(np.uint32), ref.view(np.uint32)), (single, exposure)
        q = np.floor(np.clip(ref, 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
        assert np.array_equal(b.cpu().numpy(), q)


@pytest.mark.parametrize("name,bounces,textured", [("cornell", 3, False), ("sponza_small", 6, True)])
def test_ground_truth_with_the_indirect_bounce_reenabled(oracle, hr, ctx, name, bounces, textured):
    """hr_ground_truth_params.trace_indirect = 1: the recursive traceRayEXT of rchit:95-105 un-commented (extension; the
    oracle's version is pinned to the reference's shaders with those lines un-commented by tests/test_ref_shaders.py)"""
    import torch
    from hybrid_rendering_amd import api_gi, api_post
    from oracle import pyoracle_post as opost
    W, H = 120, 72
    sd = helpers.scene_data(name)
    if textured:
        sd = synth.with_textures(sd)
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    fr = helpers.make_frames(oracle, osc, name, W, H, 1, 0.0, "soft" if name == "cornell" else "point")[0]
    sky = synth_env.sky_cubemap(8)
    env = api_gi.environment(torch.from_numpy(sky).cuda().view(torch.float16))
    o, g = opost.GroundTruthPass(W, H, max_ray_bounces=bounces, trace_indirect=True), api_post.GroundTruthPathTracer(ctx, W, H)
    g.params.max_ray_bounces, g.params.trace_indirect = bounces, 1
    base = opost.GroundTruthPass(W, H)
    for k in range(4):
        ref = o.render(osc, fr["ubo"], sky)
        g.render(gsc, fr["ubo"], env)
        torch.cuda.synchronize()
        assert np.array_equal(helpers.bits16(g.output()), ref), f"frame {k}"
        assert g.ray_count() == o.rays
    first = base.render(osc, fr["ubo"], sky)
    multi = opost.GroundTruthPass(W, H, max_ray_bounces=bounces, trace_indirect=True).render(osc, fr["ubo"], sky)
    assert oracle.f16(multi[..., :3]).mean() > oracle.f16(first[..., :3]).mean()       # the bounces add light
    g.close(); gsc.close()
