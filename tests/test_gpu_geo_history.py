"""Tolerance-mode reprojection from the pass-owned geometry records (round 4) against reprojection from the caller's
previous G-buffer (HR_GEO_HISTORY=0, what rounds 1-3 did).

The records hold verbatim copies of the G-buffer words the reprojection reads (oct normal, mesh id; for AO also the AO history value),
so every stage image must be equal BIT FOR BIT — through a history reset, a frame whose previous G-buffer arrives at other addresses
(the pass must notice and read the caller's images), a frame whose previous G-buffer aliases the current one, a switch to the parity mode and back, and a non-alternating ping_pong (AO: its
colour history then is NOT what the record holds).  reprojection.glsl:52-67,188-210."""
import os

import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env

pytestmark = pytest.mark.gpu


def _without_records(make):
    os.environ["HR_GEO_HISTORY"] = "0"
    try:
        return make()
    finally:
        del os.environ["HR_GEO_HISTORY"]


def _tables():
    import torch
    sob, sr = synth.blue_noise_tables()
    return torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()


def _frames(hr, gsc, name, W, H, n, dolly=1.5):
    cams = helpers.cameras(name, W / H, n + 1, dolly)
    light = helpers.light_for(name)
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(n)]
    return ubos, [gsc.gbuffer(u, W, H) for u in ubos]


def _clone(g):
    return {k: v.clone() for k, v in g.items()}


# frame -> what happens before it: the script every pass below runs
SCRIPT = {3: "reset", 5: "other_addresses", 6: "aliased", 7: "exact", 8: "fast_again", 10: "same_ping_pong"}
N = 12


def _sequence(hr, gsc, name, W, H, scale=0):
    ubos, gbs = _frames(hr, gsc, name, W, H, N)
    lows = [hr.gbuffer_mip(g, scale) for g in gbs] if scale else gbs
    return ubos, gbs, lows


def _drive(passes, render, images, frame_inputs, lows):
    import torch
    pp = 0
    for f in range(N):
        what = SCRIPT.get(f)
        prev = lows[f - 1] if f else lows[0]
        if what == "other_addresses":
            prev = _clone(prev)                       # same texels at other addresses: the records must NOT be trusted
        if what == "aliased":
            prev = None                                # the previous G-buffer IS the current one (one buffer rewritten in place)
        if what == "reset":
            for p in passes:
                p.reset_history()
        for p in passes:
            p.params.exact = 1 if what == "exact" else 0
        if what != "same_ping_pong":
            pp ^= 1
        fi = frame_inputs(f, prev, pp)   # prev None -> the current images
        for p in passes:
            render(p, fi, f)
        torch.cuda.synchronize()
        for label, a, b in images(passes[0], passes[1], pp):
            a, b = helpers.bits16(a) if a.dtype == torch.float16 else a.cpu().numpy(), helpers.bits16(b) if b.dtype == torch.float16 else b.cpu().numpy()
            assert np.array_equal(a, b), f"frame {f} ({what or 'plain'}): {label} differs in {(a != b).sum()} values"


@pytest.mark.parametrize("name,W,H", [("sponza_small", 320, 184), ("cornell", 203, 117)])
def test_shadows_records_equal_the_callers_gbuffer(hr, ctx, name, W, H):
    sd = helpers.scene_data(name)
    gsc = hr.Scene(ctx, sd)
    ubos, gbs, lows = _sequence(hr, gsc, name, W, H)
    sob_d, sr_d = _tables()
    a, b = hr.RayTracedShadows(ctx, W, H), _without_records(lambda: hr.RayTracedShadows(ctx, W, H))
    _drive([a, b], lambda p, fi, f: p.render(gsc, fi),
           lambda p, q, pp: [("temporal", p.image(p.IMG_TEMPORAL), q.image(q.IMG_TEMPORAL)), ("moments", p.image(p.IMG_MOMENTS1 if pp else p.IMG_MOMENTS0), q.image(q.IMG_MOMENTS1 if pp else q.IMG_MOMENTS0)),
                             ("a-trous output", p.output(hr.OUTPUT_ATROUS), q.output(hr.OUTPUT_ATROUS)), ("feedback image", p.image(p.IMG_PREV), q.image(q.IMG_PREV)),
                             ("tile classes", p.image(p.IMG_TILES), q.image(q.IMG_TILES))],
           lambda f, prev, pp: hr.frame_inputs(gbs[f], prev if prev is not None else gbs[f], ubos[f], f, pp, sob_d, sr_d), lows)
    a.close(); b.close(); gsc.close()


@pytest.mark.parametrize("name,W,H,spp", [("sponza_small", 320, 184, 4), ("cornell", 203, 117, 1)])
def test_ao_records_equal_the_callers_gbuffer(hr, ctx, name, W, H, spp):
    sd = helpers.scene_data(name)
    gsc = hr.Scene(ctx, sd)
    ubos, gbs, lows = _sequence(hr, gsc, name, W, H)
    sob_d, sr_d = _tables()
    zbp = synth.z_buffer_params()
    a, b = hr.RayTracedAO(ctx, W, H, 0), _without_records(lambda: hr.RayTracedAO(ctx, W, H, 0))
    for p in (a, b):
        p.params.spp = spp
    _drive([a, b], lambda p, fi, f: p.render(gsc, fi),
           lambda p, q, pp: [("temporal AO", p.image(p.IMG_AO1 if pp else p.IMG_AO0), q.image(q.IMG_AO1 if pp else q.IMG_AO0)),
                             ("history length", p.image(p.IMG_LEN1 if pp else p.IMG_LEN0), q.image(q.IMG_LEN1 if pp else q.IMG_LEN0)),
                             ("blurred AO", p.image(p.IMG_BLUR1), q.image(q.IMG_BLUR1)), ("tile classes", p.image(p.IMG_TILES), q.image(q.IMG_TILES))],
           lambda f, prev, pp: hr.frame_inputs(gbs[f], prev if prev is not None else gbs[f], ubos[f], f, pp, sob_d, sr_d, z_buffer_params=zbp), lows)
    a.close(); b.close(); gsc.close()


@pytest.mark.parametrize("W,H,scale", [(288, 160, 0), (333, 170, 1)])
def test_reflections_records_equal_the_callers_gbuffer(hr, ctx, W, H, scale):
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections
    name = "sponza_small"
    sd = helpers.scene_data(name)
    gsc = hr.Scene(ctx, sd)
    ubos, gbs, lows = _sequence(hr, gsc, name, W, H, scale)
    for g in gbs + (lows if scale else []):   # a roughness multiplier on the polished materials, so that the mirror regime exists
        ch = g["gb3"][..., 0]
        ch[ch == 0.1] = 0.03
    sob_d, sr_d = _tables()
    lo, hi = sd.bounds()
    u = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
    sky = synth_env.sky_cubemap(16)
    f16 = lambda t: torch.from_numpy(t).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 16, 5, f16(synth_env.brdf_lut(16)))
    ddgi = api_gi.DDGI(ctx, W, H, u)
    a, b = api_reflections.RayTracedReflections(ctx, W, H, scale), _without_records(lambda: api_reflections.RayTracedReflections(ctx, W, H, scale))
    rng = np.random.RandomState(3)

    def render(p, fi, f):
        if p is a:   # DDGI once per frame, before the first of the two reflections passes
            ddgi.render(gsc, hr.frame_inputs(gbs[f], gbs[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d), env, synth_env.random_orientation(rng))
        p.set_camera_delta((-1.5, 0.0, 0.0) if f else (0.0, 0.0, 0.0))
        p.render(gsc, fi, env, ddgi)
    _drive([a, b], render,
           lambda p, q, pp: [("temporal colour", p.image(p.IMG_COLOR1 if pp else p.IMG_COLOR0), q.image(q.IMG_COLOR1 if pp else q.IMG_COLOR0)),
                             ("moments", p.image(p.IMG_MOMENTS1 if pp else p.IMG_MOMENTS0), q.image(q.IMG_MOMENTS1 if pp else q.IMG_MOMENTS0)),
                             ("a-trous output", p.output(hr.OUTPUT_ATROUS), q.output(hr.OUTPUT_ATROUS)), ("output", p.output(hr.OUTPUT_UPSAMPLE), q.output(hr.OUTPUT_UPSAMPLE)),
                             ("tile classes", p.image(p.IMG_TILES), q.image(q.IMG_TILES))],
           lambda f, prev, pp: hr.frame_inputs(lows[f], prev if prev is not None else lows[f], ubos[f], f, pp, sob_d, sr_d, cur_full=gbs[f]), lows)
    a.close(); b.close(); ddgi.close(); gsc.close()
