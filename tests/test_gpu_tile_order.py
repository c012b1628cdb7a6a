"""Heaviest-first launch order of the trace kernels (csrc/tile_order.h, round 4) against plain blockIdx order (HR_TILE_ORDER=0).

From the second frame on a trace launch maps blockIdx through a permutation sorted by how long each tile's wave lived in the previous
frame.  Every tile is still traced exactly once by the same code, so the visibility masks, ray counts, trace images and everything the
denoisers make of them must be equal BIT FOR BIT — on a ragged image (edge tiles), on a band (fewer tile rows than the image), through
a history reset and through a statistics pass (which runs in the same order but must not disturb the recorded costs).  Variants of the
shadow pass: the sort as a launch of its own instead of riding along with the temporal kernel (HR_TILE_ORDER_FUSED=0), and the
identity list the pass falls back to when its tile costs are narrowly spread (forced with HR_TILE_ORDER_SPREAD=64)."""
import os

import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env

pytestmark = pytest.mark.gpu

N = 7


def _with_env(name, value, make):
    os.environ[name] = value
    try:
        return make()
    finally:
        del os.environ[name]


def _plain_order(make):
    return _with_env("HR_TILE_ORDER", "0", make)


def _tables():
    import torch
    sob, sr = synth.blue_noise_tables()
    return torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()


def _frames(gsc, name, W, H):
    cams = helpers.cameras(name, W / H, N + 1, 1.5)
    light = helpers.light_for(name)
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(N)]
    return ubos, [gsc.gbuffer(u, W, H) for u in ubos]


def _same(label, f, a, b):
    import torch
    a = helpers.bits16(a) if a.dtype == torch.float16 else a.cpu().numpy()
    b = helpers.bits16(b) if b.dtype == torch.float16 else b.cpu().numpy()
    assert np.array_equal(a, b), f"frame {f}: {label} differs in {(a != b).sum()} values"


@pytest.mark.parametrize("name,W,H,band", [("sponza_small", 320, 184, None), ("cornell", 203, 117, None), ("sponza_small", 320, 184, (64, 128, 16, 16))])
def test_shadows_and_ao_do_not_depend_on_the_launch_order(hr, ctx, name, W, H, band):
    import torch
    sd = helpers.scene_data(name)
    gsc = hr.Scene(ctx, sd)
    ubos, gbs = _frames(gsc, name, W, H)
    sob_d, sr_d = _tables()
    zbp = synth.z_buffer_params()
    kw = {} if band is None else {"band": band}
    sh = [hr.RayTracedShadows(ctx, W, H, **kw), _plain_order(lambda: hr.RayTracedShadows(ctx, W, H, **kw)),
          _with_env("HR_TILE_ORDER_FUSED", "0", lambda: hr.RayTracedShadows(ctx, W, H, **kw)), _with_env("HR_TILE_ORDER_SPREAD", "64", lambda: hr.RayTracedShadows(ctx, W, H, **kw))]
    ao = [hr.RayTracedAO(ctx, W, H, 0, **kw), _plain_order(lambda: hr.RayTracedAO(ctx, W, H, 0, **kw))]
    for p in sh + ao:
        p.params.exact = 0
    for p in ao:
        p.params.spp = 2
    for f in range(N):
        fi = hr.frame_inputs(gbs[f], gbs[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d, z_buffer_params=zbp)
        if f == 4:
            for p in sh + ao:
                p.reset_history()
        for p in sh + ao:
            p.render(gsc, fi)
        torch.cuda.synchronize()
        if f == 3:   # the instrumented kernel walks the same order and must leave the cost record alone
            assert sh[0].trace_stats(gsc, fi) == sh[1].trace_stats(gsc, fi) == sh[2].trace_stats(gsc, fi) == sh[3].trace_stats(gsc, fi)
            assert ao[0].trace_stats(gsc, fi) == ao[1].trace_stats(gsc, fi)
        b = sh[1]
        for a in (sh[0], sh[2], sh[3]):
            _same("shadow mask", f, a.image(a.IMG_MASK), b.image(b.IMG_MASK))
            _same("shadows a-trous output", f, a.output(hr.OUTPUT_ATROUS), b.output(hr.OUTPUT_ATROUS))
            _same("shadows moments", f, a.image(a.IMG_MOMENTS1 if f & 1 else a.IMG_MOMENTS0), b.image(b.IMG_MOMENTS1 if f & 1 else b.IMG_MOMENTS0))
            assert a.ray_count() == b.ray_count()
        a, b = ao
        _same("AO mask planes", f, a.image(a.IMG_MASK), b.image(b.IMG_MASK))
        _same("blurred AO", f, a.image(a.IMG_BLUR1), b.image(b.IMG_BLUR1))
        assert a.ray_count() == b.ray_count()
    for p in sh + ao:
        p.close()
    gsc.close()


@pytest.mark.parametrize("W,H,scale", [(288, 160, 0), (333, 170, 1)])
def test_reflections_do_not_depend_on_the_launch_order(hr, ctx, W, H, scale):
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections
    name = "sponza_small"
    sd = helpers.scene_data(name)
    gsc = hr.Scene(ctx, sd)
    ubos, gbs = _frames(gsc, name, W, H)
    lows = [hr.gbuffer_mip(g, scale) for g in gbs] if scale else gbs
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
    a, b = api_reflections.RayTracedReflections(ctx, W, H, scale), _plain_order(lambda: api_reflections.RayTracedReflections(ctx, W, H, scale))
    rng = np.random.RandomState(3)
    for f in range(N):
        ddgi.render(gsc, hr.frame_inputs(gbs[f], gbs[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d), env, synth_env.random_orientation(rng))
        fi = hr.frame_inputs(lows[f], lows[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d, cur_full=gbs[f])
        for p in (a, b):
            p.params.exact = 0
            p.set_camera_delta((-1.5, 0.0, 0.0) if f else (0.0, 0.0, 0.0))
            p.render(gsc, fi, env, ddgi)
        torch.cuda.synchronize()
        _same("trace image", f, a.image(a.IMG_TRACE), b.image(b.IMG_TRACE))
        _same("output", f, a.output(hr.OUTPUT_UPSAMPLE), b.output(hr.OUTPUT_UPSAMPLE))
        _same("tile classes", f, a.image(a.IMG_TILES), b.image(b.IMG_TILES))
        assert a.ray_count() == b.ray_count()
    a.close(); b.close(); ddgi.close(); gsc.close()


def test_launch_list_is_always_a_permutation_and_a_first_frame_capture_replays_with_it(hr, ctx):
    """VERDICT r4 weak #6 / ADVICE r4: the trace launch takes the launch list from the FIRST frame on (the identity until a sort has run), so
    (a) right after creation and after reset_history the list is a valid permutation — never the zeros of a fresh buffer;
    (b) a hipGraph captured on the very first frame (whose temporal launch carries the riding sort) re-sorts on every replay: after a few
        replays the list is a non-identity permutation and the masks equal those of eager frames."""
    import torch
    name, W, H = "sponza_small", 320, 184
    sd = helpers.scene_data(name)
    gsc = hr.Scene(ctx, sd)
    ubos, gbs = _frames(gsc, name, W, H)
    sob_d, sr_d = _tables()
    fi = hr.frame_inputs(gbs[1], gbs[0], ubos[1], 1, 1, sob_d, sr_d)
    for make in (lambda: hr.RayTracedShadows(ctx, W, H), lambda: hr.RayTracedAO(ctx, W, H, hr.SCALE_FULL_RES)):
        eager, graphed = make(), make()
        for p in (eager, graphed):
            p.params.exact = 0
            if hasattr(p.params, "spp"):
                p.params.spp = 2
        n = len(graphed.launch_order())
        ident = np.arange(n, dtype=np.uint32)
        assert n > 0 and np.array_equal(graphed.launch_order(), ident)     # (a) at creation (n = the pass's own 8x8 tiles: AO defaults to half resolution)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            graphed.render(gsc, fi)                      # the pass's FIRST frame is the captured one
        for _ in range(4):
            g.replay()
            eager.render(gsc, fi)
        torch.cuda.synchronize()
        lo = graphed.launch_order()
        assert np.array_equal(np.sort(lo), ident), "the launch list must be a permutation"
        if isinstance(graphed, hr.RayTracedAO):          # AO always sorts; the shadow pass keeps the image order when its costs are narrowly spread
            assert not np.array_equal(lo, ident), "replays of a first-frame capture never sorted"
            assert np.array_equal(np.sort(eager.launch_order()), ident) and not np.array_equal(eager.launch_order(), ident)
        _same("mask (graph replays vs eager frames)", 1, graphed.image(graphed.IMG_MASK), eager.image(eager.IMG_MASK))
        graphed.reset_history()
        assert np.array_equal(np.sort(graphed.launch_order()), ident)                                      # (a) after a reset
        eager.close(); graphed.close()
    gsc.close()
