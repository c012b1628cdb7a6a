"""Fused tolerance-mode kernels (round 3) against the kernels they fuse.

The fused launches keep the intermediate image in LDS but round it to fp16 exactly as the stored image would have been, and follow
the same per-texel rules (tile classes, sky texels, texels outside the image or outside the resident rows of a band), so their
outputs must equal the unfused chain's BIT FOR BIT — a stronger statement than the tolerance they both obey against the oracle
(tests/test_gpu_tolerance.py, which runs the fused forms since they are what render() launches):

  * kf_ao_blur_xy            == kf_ao_blur (X) then kf_ao_blur (Y)           ao_denoise_bilateral_blur.comp:75-139
  * kf_shadows_atrous01      == kf_shadows_atrous_lds<1> then <2>            shadows_denoise_atrous.comp:94-174
  * kf_refl_atrous01         == kf_refl_atrous<1> then <2>                   reflections_denoise_atrous.comp:94-181
"""
import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env, tiling

pytestmark = pytest.mark.gpu


def _tables():
    import torch
    sob, sr = synth.blue_noise_tables()
    return torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()


def _frames(hr, gsc, name, W, H, n, dolly=1.5):
    cams = helpers.cameras(name, W / H, n + 1, dolly)
    light = helpers.light_for(name)
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(n)]
    return ubos, [gsc.gbuffer(u, W, H) for u in ubos]


@pytest.mark.parametrize("name,W,H,spp,band", [("sponza_small", 320, 184, 4, None), ("sponza_small", 333, 141, 1, None), ("cornell", 250, 166, 2, None),
                                               ("sponza_small", 320, 240, 2, (80, 160))])
def test_ao_blur_xy_equals_two_passes(hr, ctx, name, W, H, spp, band):
    import torch
    sd = helpers.scene_data(name)
    gsc = hr.Scene(ctx, sd)
    ubos, gbs = _frames(hr, gsc, name, W, H, 4)
    sob_d, sr_d = _tables()
    zbp = synth.z_buffer_params()
    bnd = (band[0], band[1], tiling.HALO, tiling.HISTORY_HALO) if band else None
    fused, staged = hr.RayTracedAO(ctx, W, H, 0, band=bnd), hr.RayTracedAO(ctx, W, H, 0, band=bnd)
    for g in (fused, staged):
        g.params.spp, g.params.exact = spp, 0
    r0, r1 = (band[0], band[1]) if band else (0, H)
    for f in range(4):
        fi = hr.frame_inputs(gbs[f], gbs[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d, z_buffer_params=zbp)
        fused.render(gsc, fi)
        staged.ray_trace(gsc, fi)
        staged.temporal(fi)
        staged.blur(fi, 0)
        staged.blur(fi, 1)
        torch.cuda.synchronize()
        a, b = helpers.bits16(fused.image(fused.IMG_BLUR1))[r0:r1], helpers.bits16(staged.image(staged.IMG_BLUR1))[r0:r1]
        assert np.array_equal(a, b), f"frame {f}: fused X+Y blur differs from the two launches in {(a != b).sum()} texels"
    v = a.view(np.float16).astype(np.float32)
    assert 0.05 < (v < 0.999).mean() < 0.999, "the frame must have occluded and unoccluded texels"
    fused.close(); staged.close(); gsc.close()


def _unfused(make):
    """an instance of a pass created with HR_FUSE=0 (developer switch, read once in hr_*_create): the unfused launches"""
    import os
    os.environ["HR_FUSE"] = "0"
    try:
        return make()
    finally:
        del os.environ["HR_FUSE"]


@pytest.mark.parametrize("name,W,H,band,params", [
    ("sponza_small", 320, 184, None, None),
    ("sponza_small", 333, 141, None, dict(filter_iterations=2, power=2.0)),
    ("cornell", 250, 166, None, dict(feedback_iteration=0, phi_normal=12.5, sigma_depth=0.6)),
    ("sponza_small", 320, 240, (80, 160), None),
])
def test_shadows_atrous01_equals_two_launches(hr, ctx, name, W, H, band, params):
    import torch
    sd = helpers.scene_data(name)
    gsc = hr.Scene(ctx, sd)
    ubos, gbs = _frames(hr, gsc, name, W, H, 4)
    sob_d, sr_d = _tables()
    bnd = (band[0], band[1], tiling.HALO, tiling.HISTORY_HALO) if band else None
    fused = hr.RayTracedShadows(ctx, W, H, 0, band=bnd)
    plain = _unfused(lambda: hr.RayTracedShadows(ctx, W, H, 0, band=bnd))
    for g in (fused, plain):
        g.params.exact = 0
        for k, v in (params or {}).items():
            setattr(g.params, k, v)
    y0, y1 = (max(0, band[0] - tiling.HALO), min(H, band[1] + tiling.HALO)) if band else (0, H)   # every resident row, halo rows included
    for f in range(4):
        fi = hr.frame_inputs(gbs[f], gbs[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d)
        fused.render(gsc, fi)
        plain.render(gsc, fi)
        torch.cuda.synchronize()
        for what, a, b in (("a-trous output", fused.output(hr.OUTPUT_ATROUS), plain.output(hr.OUTPUT_ATROUS)),
                           ("feedback image", fused.image(fused.IMG_PREV), plain.image(plain.IMG_PREV))):
            a, b = helpers.bits16(a)[y0:y1], helpers.bits16(b)[y0:y1]
            assert np.array_equal(a, b), f"frame {f}: {what} of the fused iterations 0 + 1 differs in {(a != b).sum()} halfs"
    v = helpers.bits16(fused.output(hr.OUTPUT_ATROUS))[y0:y1, :, 0].view(np.float16).astype(np.float32)
    assert 0.02 < (v > 0).mean() < 0.999
    fused.close(); plain.close(); gsc.close()


@pytest.mark.parametrize("W,H,scale,band,params", [
    (288, 160, 0, None, None),
    (333, 170, 1, None, dict(approximate_with_ddgi=0, blur_as_input=1, feedback_iteration=1)),
    (224, 128, 0, None, dict(blur_as_input=1, feedback_iteration=0, filter_iterations=2, phi_normal=8.0)),
    (320, 240, 0, (80, 160), None),
])
def test_reflections_atrous01_equals_two_launches(hr, ctx, W, H, scale, band, params):
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections
    name = "sponza_small"
    sd = helpers.scene_data(name)
    gsc = hr.Scene(ctx, sd)
    ubos, gbs = _frames(hr, gsc, name, W, H, 4)
    for g in gbs:   # g_buffer.frag:106 with a roughness multiplier on the polished materials, so that the mirror regime exists
        ch = g["gb3"][..., 0]
        ch[ch == 0.1] = 0.03
    lows = [hr.gbuffer_mip(g, scale) for g in gbs] if scale else gbs
    sob_d, sr_d = _tables()
    lo, hi = sd.bounds()
    u = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
    sky = synth_env.sky_cubemap(16)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 16, 5, f16(synth_env.brdf_lut(16)))
    ddgi = api_gi.DDGI(ctx, W, H, u)
    h = H >> scale
    bnd = (band[0], band[1], tiling.HALO, tiling.HISTORY_HALO) if band else None
    fused = api_reflections.RayTracedReflections(ctx, W, H, scale, band=bnd)
    plain = _unfused(lambda: api_reflections.RayTracedReflections(ctx, W, H, scale, band=bnd))
    for g in (fused, plain):
        g.params.exact = 0
        for k, v in (params or {}).items():
            setattr(g.params, k, v)
    y0, y1 = (max(0, band[0] - tiling.HALO), min(h, band[1] + tiling.HALO)) if band else (0, h)
    rng = np.random.RandomState(3)
    for f in range(4):
        fi_full = hr.frame_inputs(gbs[f], gbs[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d)
        ddgi.render(gsc, fi_full, env, synth_env.random_orientation(rng))
        fi = hr.frame_inputs(lows[f], lows[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d, cur_full=gbs[f])
        for g in (fused, plain):
            g.set_camera_delta((-1.5, 0.0, 0.0) if f else (0.0, 0.0, 0.0))
            g.render(gsc, fi, env, ddgi)
        torch.cuda.synchronize()
        for what, a, b in (("a-trous output", fused.output(hr.OUTPUT_ATROUS), plain.output(hr.OUTPUT_ATROUS)),
                           ("feedback image", fused.image(fused.IMG_PREV), plain.image(plain.IMG_PREV))):
            a, b = helpers.bits16(a)[y0:y1], helpers.bits16(b)[y0:y1]
            assert np.array_equal(a, b), f"frame {f}: {what} of the fused iterations 0 + 1 differs in {(a != b).sum()} halfs"
    tiles = fused.image(fused.IMG_TILES).cpu().numpy()
    assert 0.02 < tiles.mean() < 1.0, "the frame must have filtered and copied tiles"
    fused.close(); plain.close(); ddgi.close(); gsc.close()
