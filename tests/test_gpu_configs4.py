"""BASELINE configs[4] at its STATED parameters against the oracle (missing #1 / #5):

 (a) DDGI 16x8x16 probes x 256 rays per probe on a 3840x2160 frame, two frames (the second with camera motion, hysteresis and
     the infinite-bounce feedback through the previous atlases) — the launch the reference records as
     vkCmdTraceRaysKHR(rays_per_probe, nProbes, 1) (ddgi.cpp:819), probe update in 4 LDS batches of 64 rays
     (gi_probe_update.glsl:58-130), border update, per-pixel sample (ddgi.cpp:896-899).  Ray images + atlases bit-exact in both
     arithmetic modes, the 4K sample bit-exact (exact = 1) / within the stated tolerance (exact = 0);
 (b) AO at 4 spp on the 3840x2160 frame (two moving frames, every stage) and on one 270-row band of the 8-GPU decomposition;
 (c) the hybrid frame in the mode bench.py times (exact = 0), at 1920x1080 and at 3840x2160 — configs[4] on one GPU end to end —
     through the deferred composite (deferred.frag:177-205) against the oracle's composite of the oracle's pass outputs: the image a
     user sees.
"""
import numpy as np
import pytest

import helpers
import test_gpu_tolerance as tol
from hybrid_rendering_amd import synth, synth_env, tiling

pytestmark = pytest.mark.gpu


def _tables():
    import torch
    sob, sr = synth.blue_noise_tables()
    return sob, sr, torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()


def _host(gb):
    import torch
    return {n: (t.cpu().numpy().view(np.uint16) if t.dtype == torch.float16 else t.cpu().numpy()) for n, t in gb.items()}


@pytest.fixture(scope="module")
def sponza_full(hr, ctx):
    sd = helpers.scene_data("sponza")
    return dict(sd=sd, scene=hr.Scene(ctx, sd))


@pytest.fixture(scope="module")
def frames4k(oracle, hr, ctx, sponza_full):
    """two consecutive 3840x2160 views (dolly 0.5 per frame) of the bench scene, on host and device"""
    W, H = 3840, 2160
    light = synth.sponza_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
    gbs_d = [sponza_full["scene"].gbuffer(u, W, H) for u in ubos]
    gbs = [_host(g) for g in gbs_d]
    return dict(W=W, H=H, ubos=ubos, gbs=gbs, gbs_d=gbs_d, osc=oracle.Scene(sponza_full["sd"]))


def test_4k_ddgi_16x8x16_256_rays_matches_oracle(oracle, hr, ctx, sponza_full, frames4k):
    import torch
    from hybrid_rendering_amd import api_gi
    from oracle import pyoracle_ddgi as od
    W, H, ubos, gbs, gbs_d, osc = (frames4k[k] for k in ("W", "H", "ubos", "gbs", "gbs_d", "osc"))
    sob, sr, sob_d, sr_d = _tables()
    gsc = sponza_full["scene"]
    lo, hi = sponza_full["sd"].bounds()
    u = synth_env.ddgi_uniforms(lo, hi, probe_counts=(16, 8, 16), rays_per_probe=256, normal_bias=0.25)
    sky = synth_env.sky_cubemap(32)
    env = api_gi.environment(torch.from_numpy(sky).cuda().view(torch.float16))
    rng = np.random.RandomState(5)
    g_ex, g_fast, o = api_gi.DDGI(ctx, W, H, u), api_gi.DDGI(ctx, W, H, u), od.DDGIPass(u)
    g_fast.params.exact = 0
    for f in range(2):
        orient = synth_env.random_orientation(rng)
        o.render(osc, ubos[f], gbs[f], sky, orient, f)
        fi = hr.frame_inputs(gbs_d[f], gbs_d[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d)
        g_ex.render(gsc, fi, env, orient)
        g_fast.render(gsc, fi, env, orient)
        torch.cuda.synchronize()
        st = o.stages
        assert st["rays"] > 1_000_000, "524,288 probe rays + the light and sky rays of their hits"
        irr, dep = o.current_read()
        for g, mode in ((g_ex, "exact = 1"), (g_fast, "exact = 0")):
            assert g.ray_count() == st["rays"], f"frame {f} ({mode}): ray count"
            rad, dd = helpers.bits16(g.image(g.IMG_RADIANCE)), helpers.bits16(g.image(g.IMG_DIRDIST))
            assert np.array_equal(rad.reshape(st["radiance"].shape), st["radiance"]), f"frame {f} ({mode}): radiance of the 2048 x 256 probe rays"
            assert np.array_equal(dd.reshape(st["direction_distance"].shape), st["direction_distance"]), f"frame {f} ({mode}): direction / hit distance"
            gi, gd = g.current_read()
            assert np.array_equal(helpers.bits16(gi), irr), f"frame {f} ({mode}): irradiance atlas (256 rays = 4 LDS batches)"
            assert np.array_equal(helpers.bits16(gd), dep), f"frame {f} ({mode}): depth atlas"
        out = helpers.bits16(g_ex.output())
        assert out.shape[:2] == (H, W)
        assert np.array_equal(out, st["output"]), f"frame {f}: 4K probe-grid sample differs in {(out != st['output']).sum()} halfs"
        tol.compare16(helpers.bits16(g_fast.output()), st["output"], f"frame {f}: 4K probe-grid sample (exact = 0), all channels", outlier_pixels=tol.DDGI_OUTLIERS)
    assert oracle.f16(irr).max() > 0.05
    g_ex.close(); g_fast.close()


def test_4k_ao_4spp_matches_oracle(oracle, hr, ctx, sponza_full, frames4k):
    """configs[4]'s AO: 4 spp at 3840x2160, two moving frames, whole frame in both modes + rank 4's band of the 8-GPU cut (frame 0)"""
    import torch
    W, H, ubos, gbs, gbs_d, osc = (frames4k[k] for k in ("W", "H", "ubos", "gbs", "gbs_d", "osc"))
    sob, sr, sob_d, sr_d = _tables()
    gsc = sponza_full["scene"]
    zbp = synth.z_buffer_params()
    spp = 4
    o = oracle.AOPass(W, H, spp=spp, zbp=zbp)
    g_ex, g_fast = hr.RayTracedAO(ctx, W, H, 0), hr.RayTracedAO(ctx, W, H, 0)
    b0, b1 = tiling.band_rows(H, 8, 4)
    band = (b0, b1, tiling.HALO, tiling.HISTORY_HALO)
    g_band, g_band_fast = hr.RayTracedAO(ctx, W, H, 0, band=band), hr.RayTracedAO(ctx, W, H, 0, band=band)
    for g in (g_ex, g_fast, g_band, g_band_fast):
        g.params.spp = spp
    g_fast.params.exact = g_band_fast.params.exact = 0
    mh = (H + 3) // 4
    for f in range(2):
        cur, prev = gbs[f], gbs[f - 1 if f else 0]
        o.render(osc, ubos[f], cur, prev, sob, sr, f)
        st = o.stages
        fi = hr.frame_inputs(gbs_d[f], gbs_d[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d, z_buffer_params=zbp)
        g_ex.render(gsc, fi)
        g_fast.render(gsc, fi)
        torch.cuda.synchronize()
        assert st["rays"] > 4 * 6_000_000
        for g, mode in ((g_ex, "exact = 1"), (g_fast, "exact = 0")):
            mask = g.image(g.IMG_MASK).cpu().numpy().view(np.uint32)[:spp * mh].reshape(spp, mh, -1)
            assert np.array_equal(mask, st["mask"]), f"frame {f} ({mode}): the four mask planes"
            assert g.ray_count() == st["rays"]
        assert np.array_equal(g_ex.image(g_ex.IMG_TILES).cpu().numpy(), st["tiles"]), f"frame {f}: tile classes"
        assert np.array_equal(helpers.bits16(g_ex.image(g_ex.IMG_AO1 if f & 1 else g_ex.IMG_AO0)), st["temporal"]), f"frame {f}: temporal AO"
        assert np.array_equal(helpers.bits16(g_ex.image(g_ex.IMG_LEN1 if f & 1 else g_ex.IMG_LEN0)), st["length"]), f"frame {f}: history length"
        assert np.array_equal(helpers.bits16(g_ex.image(g_ex.IMG_BLUR0)), st["blur0"]), f"frame {f}: blur X"
        assert np.array_equal(helpers.bits16(g_ex.image(g_ex.IMG_BLUR1)), st["blur1"]), f"frame {f}: blur Y"
        ex = tol.tiles_close(g_fast.image(g_fast.IMG_TILES).cpu().numpy(), st["tiles"], f"frame {f} (exact = 0)", shape=(H, W))
        tol.compare16(helpers.bits16(g_fast.image(g_fast.IMG_AO1 if f & 1 else g_fast.IMG_AO0)), st["temporal"], f"frame {f} temporal AO (exact = 0)")
        assert np.array_equal(helpers.bits16(g_fast.image(g_fast.IMG_LEN1 if f & 1 else g_fast.IMG_LEN0)), st["length"]), f"frame {f}: history length (exact = 0) is an integer count"
        tol.compare16(helpers.bits16(g_fast.image(g_fast.IMG_BLUR1)), st["blur1"], f"frame {f} blurred AO (exact = 0)", exclude=ex)
        if f == 0:
            g_band.render(gsc, fi)
            g_band_fast.render(gsc, fi)
            torch.cuda.synchronize()
            assert np.array_equal(helpers.bits16(g_band.image(g_band.IMG_BLUR1))[b0:b1], st["blur1"][b0:b1]), "4 spp AO band rows [1080, 1344)"
            tol.compare16(helpers.bits16(g_band_fast.image(g_band_fast.IMG_BLUR1))[b0:b1], st["blur1"][b0:b1], "4 spp AO band (exact = 0)", exclude=ex[b0:b1])
    for g in (g_ex, g_fast, g_band, g_band_fast):
        g.close()


@pytest.mark.parametrize("W,H", [(1920, 1080), (3840, 2160)])
def test_hybrid_frame_tolerance_mode_composite(oracle, hr, ctx, sponza_full, W, H):
    """the frame bench.py's `passes.hybrid_1080p` / `hybrid_4k_one_gpu` times (shadows + AO 4 spp + DDGI 16x8x16x256 + half-res reflections, exact = 0) through
    k_deferred, two moving frames, against the oracle's deferred composite of the oracle's pass outputs (deferred.frag:177-205).
    Every pass output AND the final HDR image obey the image rule of DESIGN.md §3.6 (2 fp16 ulp on >= 99.9 %, rel-L2 <= 1e-3) — all channels."""
    import torch
    from hybrid_rendering_amd import api_deferred, api_gi, api_reflections
    from oracle import pyoracle_ddgi as od, pyoracle_deferred as odf, pyoracle_reflections as orf
    sd, gsc = sponza_full["sd"], sponza_full["scene"]
    osc = oracle.Scene(sd)
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(16, 8, 16), rays_per_probe=256, normal_bias=0.1)
    sky = synth_env.sky_cubemap(32)
    pre, lut, sh9 = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(32), synth_env.sh9_from_cubemap(sky)
    env_np = dict(sky=sky, prefiltered=pre, pre_size=32, pre_levels=5, lut=lut)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(pre), 32, 5, f16(lut))
    light = synth.sponza_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
    fulls = [_host(gsc.gbuffer(u, W, H)) for u in ubos]
    halves = [helpers.nearest_mip(g, 1) for g in fulls]
    sob, sr, sob_d, sr_d = _tables()
    zbp = synth.z_buffer_params()
    g_sh, g_ao = hr.RayTracedShadows(ctx, W, H), hr.RayTracedAO(ctx, W, H, 0)
    g_gi, g_rf = api_gi.DDGI(ctx, W, H, ddgi), api_reflections.RayTracedReflections(ctx, W, H, 1)
    g_ao.params.spp = 4
    for g in (g_sh, g_ao, g_gi, g_rf):
        g.params.exact = 0
    g_df = api_deferred.DeferredShading(ctx, W, H)
    g_df.set_sh9(sh9)
    o_sh, o_ao = oracle.ShadowsPass(W, H), oracle.AOPass(W, H, spp=4, zbp=zbp)
    o_gi, o_rf = od.DDGIPass(ddgi), orf.ReflectionsPass(W // 2, H // 2)
    rng = np.random.RandomState(11)
    for f in range(2):
        full, pfull, half, phalf = fulls[f], fulls[f - 1 if f else 0], halves[f], halves[f - 1 if f else 0]
        ubo, orient = ubos[f], synth_env.random_orientation(rng)
        delta = (-0.5, 0.0, 0.0) if f else (0.0, 0.0, 0.0)
        sh = o_sh.render(osc, ubo, full, pfull, sob, sr, f)
        ao = o_ao.render(osc, ubo, full, pfull, sob, sr, f)
        gi = o_gi.render(osc, ubo, full, sky, orient, f)
        irr, dep = o_gi.current_read()
        rf = o_rf.render(osc, ubo, ddgi, half, phalf, sob, sr, f, env_np, irr, dep, camera_delta=delta, full=full, ping_pong=bool(f & 1))
        ref = odf.shade(ubo, full, sh, ao, rf, gi, 0b1111, sh9, env_np)
        full_d, pfull_d, half_d, phalf_d = (helpers.to_cuda(g) for g in (full, pfull, half, phalf))
        fi_full = hr.frame_inputs(full_d, pfull_d, ubo, f, f & 1, sob_d, sr_d, z_buffer_params=zbp)
        fi_half = hr.frame_inputs(half_d, phalf_d, ubo, f, f & 1, sob_d, sr_d, cur_full=full_d, z_buffer_params=zbp)
        g_sh.render(gsc, fi_full)
        g_ao.render(gsc, fi_full)
        g_gi.render(gsc, fi_full, env, orient)
        g_rf.set_camera_delta(delta)
        g_rf.render(gsc, fi_half, env, g_gi)
        g_df.render(fi_full, env, shadow=g_sh.output(hr.OUTPUT_UPSAMPLE), ao=g_ao.output(hr.OUTPUT_UPSAMPLE),
                    reflections=g_rf.output(hr.OUTPUT_UPSAMPLE), gi=g_gi.output())
        torch.cuda.synchronize()
        ex_sh = tol.tiles_close(g_sh.image(g_sh.IMG_TILES).cpu().numpy(), o_sh.stages["tiles"], f"frame {f} shadows", shape=(H, W))
        ex_ao = tol.tiles_close(g_ao.image(g_ao.IMG_TILES).cpu().numpy(), o_ao.stages["tiles"], f"frame {f} AO", shape=(H, W))
        ex_rf = tol.upscale_mask(tol.tiles_close(g_rf.image(g_rf.IMG_TILES).cpu().numpy(), o_rf.stages["tiles"], f"frame {f} reflections", shape=(H // 2, W // 2)), 1, H, W)
        tol.compare16(helpers.bits16(g_sh.output(hr.OUTPUT_UPSAMPLE)), sh, f"frame {f} shadows output", exclude=ex_sh, variance_channels=(1,))
        tol.compare16(helpers.bits16(g_ao.output(hr.OUTPUT_UPSAMPLE)), ao, f"frame {f} AO output", exclude=ex_ao)
        tol.compare16(helpers.bits16(g_gi.output()), gi, f"frame {f} DDGI sample", outlier_pixels=tol.DDGI_OUTLIERS)
        tol.compare16(helpers.bits16(g_rf.output(hr.OUTPUT_UPSAMPLE)), rf, f"frame {f} reflections output", exclude=ex_rf, variance_channels=(3,), outlier_scale=tol.upsample_scale(1), outlier_pixels=tol.REFL_OUTLIERS)
        got = helpers.bits16(g_df.output())
        # the composite of four images that are each within 2 ulp obeys the SAME image rule (measured: it also holds at 2 ulp, not only at 4)
        tol.compare16(got, ref, f"frame {f} final HDR image", exclude=ex_sh | ex_ao | ex_rf, outlier_pixels=tol.REFL_OUTLIERS)
    img = oracle.f16(ref[..., :3])
    assert np.isfinite(img).all() and img.mean() > 0.01
    for g in (g_sh, g_ao, g_gi, g_rf, g_df):
        g.close()


def test_8k_shadows_two_frames_match_oracle(oracle, hr, ctx, sponza_full):
    """maximum size: 7680x4320 (33 M pixels, 4x configs[4]; image offsets, mask words and tile counts well past 2^24) — two moving
    frames of the whole shadow pass against the oracle: masks / ray counts / tile classes bit-exact in both modes, every stage image
    bit-exact (exact = 1) or within the stated tolerance (exact = 0); plus rank 13 of a 16-way cut as a band"""
    import torch
    W, H = 7680, 4320
    light = synth.sponza_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
    gbs_d = [sponza_full["scene"].gbuffer(u, W, H) for u in ubos]
    gbs = [_host(g) for g in gbs_d]
    osc = oracle.Scene(sponza_full["sd"])
    sob, sr, sob_d, sr_d = _tables()
    gsc = sponza_full["scene"]
    o = oracle.ShadowsPass(W, H)
    g_ex, g_fast = hr.RayTracedShadows(ctx, W, H), hr.RayTracedShadows(ctx, W, H)
    g_fast.params.exact = 0
    b0, b1 = tiling.band_rows(H, 16, 13)
    g_band = hr.RayTracedShadows(ctx, W, H, 0, band=(b0, b1, tiling.HALO, tiling.HISTORY_HALO))
    for f in range(2):
        cur, prev = gbs[f], gbs[f - 1 if f else 0]
        o.render(osc, ubos[f], cur, prev, sob, sr, f)
        st = o.stages
        fi = hr.frame_inputs(gbs_d[f], gbs_d[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d)
        g_ex.render(gsc, fi)
        g_fast.render(gsc, fi)
        torch.cuda.synchronize()
        assert st["rays"] > 8_000_000
        for g, mode in ((g_ex, "exact = 1"), (g_fast, "exact = 0")):
            assert np.array_equal(g.image(g.IMG_MASK).cpu().numpy().view(np.uint32), st["mask"]), f"frame {f} ({mode}): visibility mask"
            assert g.ray_count() == st["rays"], f"frame {f} ({mode}): ray count"
        assert np.array_equal(g_ex.image(g_ex.IMG_TILES).cpu().numpy(), st["tiles"]), f"frame {f}: tile classes"
        assert np.array_equal(helpers.bits16(g_ex.image(g_ex.IMG_TEMPORAL)), st["temporal"]), f"frame {f}: temporal"
        assert np.array_equal(helpers.bits16(g_ex.image(g_ex.IMG_MOMENTS1 if f & 1 else g_ex.IMG_MOMENTS0)), st["moments"]), f"frame {f}: moments"
        assert np.array_equal(helpers.bits16(g_ex.output(hr.OUTPUT_ATROUS)), st["output"]), f"frame {f}: a-trous output"
        ex = tol.tiles_close(g_fast.image(g_fast.IMG_TILES).cpu().numpy(), st["tiles"], f"frame {f} (exact = 0)", shape=(H, W))
        tol.compare16(helpers.bits16(g_fast.output(hr.OUTPUT_ATROUS)), st["output"], f"frame {f} a-trous output (exact = 0)", exclude=ex, variance_channels=(1,))
        if f == 0:
            g_band.render(gsc, fi)
            torch.cuda.synchronize()
            assert np.array_equal(helpers.bits16(g_band.output(hr.OUTPUT_ATROUS))[b0:b1], st["output"][b0:b1]), f"band rows [{b0}, {b1}) of the 16-way cut"
    for g in (g_ex, g_fast, g_band):
        g.close()
