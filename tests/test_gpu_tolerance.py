"""Tolerance mode (hr_*_params.exact = 0: hardware rcp / rsq / sqrt / exp / log, fused multiply-adds, re-associated sums —
csrc/denoise_fast.hip) against the CPU oracle.  THE CONTRACT these runners enforce is stated once, clause by clause, in docs/TOLERANCE.md
(north_star "AO / reflections / GI within a stated fp32 tolerance"): masks, ray counts, DDGI atlases and the reflections'
trace image bit-exact; every other fp16 image <= 2 fp16 ulp (or 2^-20 absolute) on >= 99.9 % of the texels, relative L2 <= 1e-3 (<= 1e-2 over all texels), a hard
cap of 32 ulp / 2^-10 per texel outside flipped-tile neighbourhoods, one bounded allowance for the reflections' denoised images, tile classes
equal on >= 99.5 % of the tiles.  Every threshold below is a constant (tests/test_tolerance_rule.py).  The runs are several frames long with
a moving camera, so the bound holds through the temporal feedback loops."""
import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env

pytestmark = pytest.mark.gpu


VARIANCE_FLOOR = 1e-4     # absolute slack of the variance channels (see compare16)
import os
# absolute slack of the temporal stages' INTERMEDIATE images (round 3: 2e-4; rounds 1-2 allowed 1e-3.  Everything — the tests, the 1080p frames and
# 295 random configurations of tools/fuzz_tolerance.py incl. 20-frame sequences — also passes at 1e-4; 2e-4 leaves a factor of two)
INTERMEDIATE_FLOOR = 2e-4
# Absolute floor of EVERY comparison (late round 6): two values that differ by at most 2^-20 count as equal whatever their magnitude — below 2^-10 (0.1 % of white)
# the rule is absolute.  An fp16 ulp shrinks with the value (6e-8 at 1e-4), and a stage with gain > 1 on the relative error turns a tolerated 2-ulp input into 3-5 ulp
# there: the AO upsample raises to `power` 1.2 (ao_upsample.comp), so blurred AO of 7e-4 inside its 2-ulp bound came out 3-5 ulp (4e-7) apart on 84 texels of a
# 252 x 159 quarter-resolution frame 0 (tools/fuzz_tolerance.py 6301 #279, the only miss of 1200 new sequences; docs/EXPERIMENTS.md R6.11) — a 4000th of an
# 8-bit display step.  The configuration is part of the suite: test_ao_fuzz_sequence_in_the_dark.
OUTPUT_FLOOR = 2.0 ** -20


def _key(bits):
    """fp16 bit patterns -> integers ordered like the values (so that |key_a - key_b| is the distance in fp16 ulp)"""
    b = bits.astype(np.int32)
    mag = b & 0x7fff
    return np.where(b & 0x8000, -mag, mag)


CAP_ULPS = 32     # hard per-texel cap (every texel outside flipped-tile neighbourhoods) ...
CAP_ABS = 2.0 ** -10   # ... OR this absolute difference
REPORT = os.environ.get("HR_TEST_TOLERANCE_REPORT")          # print the achieved figures of every image (the ONLY environment switch: every threshold is a constant,
                                                             # tests/test_tolerance_rule.py test_thresholds_are_constants)
# Counted allowance of pixels beyond the hard cap (round 5: only the reflections' denoised images still have one, and it is bounded tightly):
#   AO, DDGI probe-grid sample, reflections trace image, the shadows' temporal stage and every a-trous / upsample LAUNCH on its own: NONE.  Their discrete decisions are taken with the parity kernels' arithmetic
#     wherever the fast operands cannot be trusted (history taps on a knife edge of the validity test: Reproj::exact_bits; DDGI gathers whose
#     weights are ill-conditioned or NaN-driven: ddgi_sample_fast.h redo).
#   reflections temporal / moments / a-trous / upsampled output, and (late round 6) the shadows' a-trous output + feedback image END TO END: at most max(4, 2e-5 of the pixels) pixels per image (x 5 * 4^scale for an
#     upsampled output), each within OUTLIER_ULPS fp16 ulp or OUTLIER_ABS of the oracle — NOT the channel's value range of round 4.  Cause, shown
#     by tools/refl_outlier_probe.py on frame 0 of two fuzz configurations (trace and temporal images bit-identical, variance channel 0): the
#     reference's luminance edge-stopping weight is exp(-|dl| / (phi_color sqrt(1e-10 + var))) (reflections_denoise_atrous.comp:113-125,
#     edge_stopping.glsl:31-62): with var == 0 (first frames, disocclusions) the exponent changes by ~0.6 per fp16 ulp of an input texel, so ONE
#     fp16 ulp of difference in the stored intermediate of a-trous iteration i (inside the 2-ulp rule) re-weights a tap of iteration i + 1 by
#     e^0.6 — the reference's own filter is ill-conditioned there, any arithmetic that is not bit-identical meets it (measured over 200 random
#     configurations: 9 images, 1-4 texels each, <= 134 ulp / 1.1e-2).  (Until late round 6 the reflections' temporal kernel also kept the fast history-tap test; it now re-runs
#     knife-edge pixels with the parity verdicts like the other two.)
OUTLIER_PIXELS = 0.0
DDGI_OUTLIERS = 0.0
REFL_OUTLIERS = 2e-5
ATROUS_OUTLIERS = REFL_OUTLIERS   # the same allowance for the END-TO-END images of the shadows' a-trous chain (late round 6, see test_shadows_tolerance); its kernels, launch by launch, have none
OUTLIER_ULPS = 512
OUTLIER_ABS = 2.0 ** -5


def compare16(got, ref, what, rel_l2=1e-3, ulps=2, frac=0.999, abs_floor=OUTPUT_FLOOR, exclude=None, variance_channels=(), variance_floor=VARIANCE_FLOOR,
              cap_ulps=None, cap_abs=None, outlier_pixels=None, outlier_scale=1):
    """got / ref: uint16 fp16 bit patterns, ALL channels of the image.  abs_floor: differences below it count as equal (OUTPUT_FLOOR for every image;
    INTERMEDIATE_FLOOR for intermediate images whose small values are differences of nearly equal numbers, e.g. variance = m2 - m1^2).  exclude: bool [H, W] of texels
    left out of the per-texel bound (neighbourhoods of tiles whose classification differs — a discrete decision; they stay in the L2
    bound).  variance_channels: channels that carry a variance estimate (shadows .y, reflections .a): they descend from
    `m2 - m1^2` / `E[x^2] - E[x]^2`, a difference of nearly equal numbers, so their rule is "2 fp16 ulp OR |diff| <= variance_floor"
    — the quantity the next a-trous iteration reads them for is phi * sqrt(variance) (shadows_denoise_atrous.comp:65-88), on which
    a 1e-4 absolute slip is far below the 2-ulp bound of the filtered channel; the L2 bounds cover them like every other channel."""
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    outlier_pixels = OUTLIER_PIXELS if outlier_pixels is None else outlier_pixels
    g, r = got.view(np.float16).astype(np.float64), ref.view(np.float16).astype(np.float64)
    assert np.isfinite(g).all(), f"{what}: non-finite values"
    num, den = np.sqrt(((g - r) ** 2).sum()), np.sqrt((r ** 2).sum())
    rl2 = num / den if den > 0 else num
    ok = (np.abs(_key(got) - _key(ref)) <= ulps) | (np.abs(g - r) <= abs_floor)
    for c in variance_channels:
        ok[..., c] |= np.abs(g - r)[..., c] <= variance_floor
    if exclude is not None:
        ex = exclude if ok.ndim == 2 else exclude[..., None]
        ok = ok | ex
    # hard caps: no texel may be arbitrarily wrong
    cap_ulps = CAP_ULPS if cap_ulps is None else cap_ulps
    cap_abs = max(CAP_ABS if cap_abs is None else cap_abs, abs_floor)
    ulp_d, abs_d = np.abs(_key(got) - _key(ref)), np.abs(g - r)
    capped = (ulp_d <= cap_ulps) | (abs_d <= cap_abs)
    for c in variance_channels:
        capped[..., c] |= abs_d[..., c] <= max(cap_abs, variance_floor)
    exb = None if exclude is None else (exclude if ok.ndim == 2 else np.broadcast_to(exclude[..., None], ok.shape))
    outside = capped if exb is None else (capped | exb)
    if REPORT:
        sel_o = np.ones(ok.shape, bool) if exb is None else ~exb
        print(f"[tolerance] {what}: {ok.mean() * 100:.4f} % within {ulps} ulp, max {int(ulp_d[sel_o].max()) if sel_o.any() else 0} ulp / {abs_d[sel_o].max() if sel_o.any() else 0:.3e} abs outside flipped-tile "
              f"neighbourhoods ({0 if exb is None else int(exb.sum())} texels inside), rel-L2 {rl2:.2e}", flush=True)
    n_pixels = int(np.prod(ok.shape[:2]))
    # outlier_scale: an UPSAMPLED output spreads one low-resolution outlier over the full-resolution pixels whose 4-tap cross reads it —
    # its own (2^scale)^2 block and the four neighbouring blocks: 5 * 4^scale pixels (upsample_scale() below)
    allowed = int(max(4, outlier_pixels * n_pixels) * outlier_scale) if outlier_pixels > 0 else 0
    beyond_px = (~outside).reshape(ok.shape[0], ok.shape[1], -1).any(axis=2)
    if allowed and 0 < beyond_px.sum() <= allowed:
        # the counted allowance: these pixels answer to the outlier bound (OUTLIER_ULPS fp16 ulp or OUTLIER_ABS), not to the hard cap
        wild = ~outside & ~((ulp_d <= OUTLIER_ULPS) | (abs_d <= OUTLIER_ABS))
        assert not wild.any(), (f"{what}: an outlier pixel differs by more than {OUTLIER_ULPS} fp16 ulp and {OUTLIER_ABS:.2e} "
                                f"(worst {int(ulp_d[wild].max())} ulp / {abs_d[wild].max():.3e})")
        if REPORT:
            print(f"[tolerance] {what}: {int(beyond_px.sum())} pixels beyond the hard cap, inside the allowance of {allowed}", flush=True)
    elif not outside.all():
        w = np.argwhere(~outside)
        raise AssertionError(f"{what}: {len(w)} texels beyond the hard cap ({cap_ulps} fp16 ulp or {cap_abs:.2e}); worst {int(ulp_d[~outside].max())} ulp / {abs_d[~outside].max():.3e}; "
                             f"first (y, x, ...): {w[:6].tolist()} got {g[tuple(w[:6].T)].tolist()} ref {r[tuple(w[:6].T)].tolist()}")
    if exb is not None and exb.any():
        # inside the neighbourhood of a flipped tile: bounded by the value range of the channel (a copied / cleared tile swaps values, it does not invent them)
        rng = (r.max(axis=tuple(range(r.ndim - 1)) if r.ndim == 3 else None) - r.min(axis=tuple(range(r.ndim - 1)) if r.ndim == 3 else None)) + cap_abs
        over = (abs_d > rng) & exb
        assert not over.any(), f"{what}: {int(over.sum())} texels next to a flipped tile differ by more than the channel's value range (max {abs_d[exb].max():.3e})"
    f = ok.mean()
    bad = np.argwhere(~ok)
    where = f"; first offenders (y, x, ...): {bad[:6].tolist()} got {g[tuple(bad[:6].T)].tolist()} ref {r[tuple(bad[:6].T)].tolist()}" if len(bad) else ""
    sel = ok if exclude is None else (ok & ~(exclude if ok.ndim == 2 else np.broadcast_to(exclude[..., None], ok.shape)))
    rl2_in = np.sqrt(((g - r)[sel] ** 2).sum()) / max(np.sqrt((r[sel] ** 2).sum()), 1e-30)
    assert rl2_in <= rel_l2, f"{what}: relative L2 error over the texels inside the ulp bound {rl2_in:.2e} > {rel_l2:.0e}"
    # over ALL texels (round 5: the counted pixels included again — bounded by OUTLIER_ABS they cannot outweigh even a 70x32 image)
    assert rl2 <= 10 * rel_l2, f"{what}: relative L2 error over all texels {rl2:.2e} > {10 * rel_l2:.0e}"
    assert f >= frac, f"{what}: only {f * 100:.3f} % of the texels within {ulps} fp16 ulp ({len(bad)} outside, max abs diff {np.abs(g - r).max():.3e}){where}"
    return rl2, f


def compare_trace(got, ref, what):
    """the reflections' ray-trace image: BIT-EXACT in tolerance mode too, all four channels (round 6: the DDGI irradiance gathers of the hit
    shading and of the rough pixels run the parity arithmetic in both modes — one fp16 ulp of a stored colour became 0.4-100 % of a small
    variance `m2 - m1^2` one frame later, docs/EXPERIMENTS.md R5.8 / R6.1)"""
    assert np.array_equal(got[..., 3], ref[..., 3]), f"{what}: ray lengths (channel a) must be bit-exact — the traversal has one mode"
    assert np.array_equal(got, ref), f"{what}: {int((got != ref).sum())} colour values differ from the oracle's (the trace image is bit-exact in both modes)"


def upsample_scale(scale):
    """how many full-resolution pixels of an upsampled output one low-resolution texel reaches (*_upsample.comp: 4 taps at +-1 coarse texel)"""
    return 5 * 4 ** scale if scale else 1


def tiles_close(got, ref, what, frac=0.995, shape=None, reach=2):
    """tile classes agree on >= frac of the tiles; returns the per-texel exclusion mask: texels within `reach` tiles of a tile
    whose class differs (a flipped tile is copied / cleared instead of filtered, and the a-trous taps of its neighbours see it)"""
    diff = got != ref
    f = 1.0 - diff.mean()
    assert f >= frac, f"{what}: tile classes agree on {f * 100:.2f} % of the tiles"
    if shape is None:
        return None
    d = diff.copy()
    for _ in range(reach):
        p = np.pad(d, 1)
        d = p[:-2, :-2] | p[:-2, 1:-1] | p[:-2, 2:] | p[1:-1, :-2] | p[1:-1, 1:-1] | p[1:-1, 2:] | p[2:, :-2] | p[2:, 1:-1] | p[2:, 2:]
    return np.kron(d, np.ones((8, 8), bool))[:shape[0], :shape[1]]


def upscale_mask(ex, scale, H, W):
    """low-resolution exclusion mask -> full resolution (odd sizes: the last row / column repeats)"""
    up = np.kron(ex, np.ones((1 << scale, 1 << scale), bool))[:H, :W]
    return np.pad(up, ((0, H - up.shape[0]), (0, W - up.shape[1])), mode="edge")


def _tables():
    import torch
    sob, sr = synth.blue_noise_tables()
    return sob, sr, torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()


@pytest.mark.parametrize("name,w,h,dolly,light,params", [
    ("sponza_small", 320, 184, 1.5, "default", None),
    ("cornell", 250, 166, 1.0, "soft", None),
    ("sponza_small", 203, 117, 2.0, "spot", dict(filter_iterations=5, feedback_iteration=0, phi_normal=12.5, power=2.0, alpha=0.05, radius=2, sigma_depth=0.6)),
    # the hard tier's geometry (layered fabric, foliage cards) under its grazing sun: far more history taps sit near a validity threshold
    ("sponza_hard_small", 480, 270, 1.0, "grazing", None),
])
def test_shadows_tolerance(oracle, hr, ctx, name, w, h, dolly, light, params, n_frames=6):
    import torch
    sd = helpers.scene_data(name)
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    n = n_frames
    frames = helpers.make_frames(oracle, osc, name, w, h, n, dolly, light)
    sob, sr, sob_d, sr_d = _tables()
    kw = dict(params or {})
    gp, op = hr.RayTracedShadows(ctx, w, h), oracle.ShadowsPass(w, h, **kw)
    for k, v in kw.items():
        setattr(gp.params, k, v)
    gp.params.exact = 0
    for f in range(n):
        cur, prev = frames[f]["gb"], frames[f - 1 if f else 0]["gb"]
        op.render(osc, frames[f]["ubo"], cur, prev, sob, sr, f)
        gp.render(gsc, hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d))
        torch.cuda.synchronize()
        st = op.stages
        assert np.array_equal(gp.image(gp.IMG_MASK).cpu().numpy().view(np.uint32), st["mask"]), f"frame {f}: mask must be bit-exact"
        assert gp.ray_count() == st["rays"]
        ex = tiles_close(gp.image(gp.IMG_TILES).cpu().numpy(), st["tiles"], f"frame {f}", shape=(h, w))
        # intermediate images: the variance / second-moment channels are differences of nearly equal numbers
        compare16(helpers.bits16(gp.image(gp.IMG_TEMPORAL)), st["temporal"], f"frame {f} temporal", abs_floor=INTERMEDIATE_FLOOR)
        compare16(helpers.bits16(gp.image(gp.IMG_MOMENTS1 if f & 1 else gp.IMG_MOMENTS0)), st["moments"], f"frame {f} moments (m1, m2, history length, 0)", abs_floor=INTERMEDIATE_FLOOR)
        out, ref = helpers.bits16(gp.output(hr.OUTPUT_ATROUS)), st["output"]
        # Stage-wise, no allowance at all, ITERATION BY ITERATION (as for the reflections below): each a-trous launch re-run on its own (hr_shadows_atrous_iteration —
        # bit-identical to the fused launches of render(), asserted on the last image) against the ORACLE's iteration on the very image the GPU iteration read.
        # The reference's visibility weight exp(-|dv| / (phi sqrt(1e-10 + var))) (shadows_denoise_atrous.comp:65-88) is ill-conditioned where the variance is ~0:
        # with phi = 10 a ONE-ulp difference between two neighbours that are equal in the other arithmetic (4.9e-4 at 0.86) moves a tap's weight from 1 to 0.0075.
        # Over the CHAIN that is the reference's own filter spreading a tolerated ulp (fuzz 6311 #915, frame 4: temporal images 1 ulp apart, ONE texel of the
        # fifth iteration's output 51 ulp / 2.5e-2 off — the first such texel in ~5000 fuzzed shadow sequences; docs/EXPERIMENTS.md R6.12); per iteration
        # nothing is left to amplify.
        tiles_np = np.ascontiguousarray(gp.image(gp.IMG_TILES).cpu().numpy().astype(st["tiles"].dtype))
        fi_cur = hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d)
        src = helpers.bits16(gp.image(gp.IMG_TEMPORAL)).reshape(st["temporal"].shape)
        n_it = op.p["filter_iterations"]
        for i in range(n_it):
            gp.atrous_iteration(fi_cur, i)
            torch.cuda.synchronize()
            got_i = helpers.bits16(gp.image(gp.IMG_ATROUS0 if i & 1 else gp.IMG_ATROUS1)).reshape(src.shape)
            ref_i = oracle.shadows_atrous(src, cur["gb2"], cur["gb3"], tiles_np, 1 << i, op.p["radius"], op.p["phi_visibility"], op.p["phi_normal"], op.p["sigma_depth"],
                                          op.p["power"] if i == n_it - 1 else 0.0)
            compare16(got_i, ref_i, f"frame {f} a-trous iteration {i} against the oracle's iteration on the same input image", variance_channels=(1,))
            src = got_i
        assert np.array_equal(src, out.reshape(src.shape)), f"frame {f}: the iterations one by one must reproduce render()'s a-trous output bit for bit"
        # the chain end to end: the image rule + the counted allowance of the a-trous chains (ATROUS_OUTLIERS: at most max(4, 2e-5 of the pixels) pixels beyond the
        # cap, each within OUTLIER_ULPS / OUTLIER_ABS)
        compare16(out, ref, f"frame {f} denoised visibility + filtered variance", exclude=ex, variance_channels=(1,), outlier_pixels=ATROUS_OUTLIERS)
        compare16(helpers.bits16(gp.image(gp.IMG_PREV)), op.prev_image, f"frame {f} feedback image (next frame's history)", exclude=ex, variance_channels=(1,), outlier_pixels=ATROUS_OUTLIERS)
    gp.close(); gsc.close()


def test_shadows_half_res_upsample_tolerance(oracle, hr, ctx):
    import torch
    name, W, H, scale = "sponza_small", 320, 176, 1
    sd = helpers.scene_data(name)
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    frames = helpers.make_frames(oracle, osc, name, W, H, 4, 1.0, scale_mips=scale)
    sob, sr, sob_d, sr_d = _tables()
    w, h = W >> scale, H >> scale
    gp, op = hr.RayTracedShadows(ctx, W, H, scale), oracle.ShadowsPass(w, h)
    gp.params.exact = 0
    for f in range(4):
        cur, prev, full = frames[f]["mips"][scale], frames[f - 1 if f else 0]["mips"][scale], frames[f]["gb"]
        op.render(osc, frames[f]["ubo"], cur, prev, sob, sr, f)
        up = oracle.upsample(full, cur, op.stages["output"], channels=1, sky_value=0.0, power=0.0)[..., 0]
        gp.render(gsc, hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d, cur_full=helpers.to_cuda(full)))
        torch.cuda.synchronize()
        assert np.array_equal(gp.image(gp.IMG_MASK).cpu().numpy().view(np.uint32), op.stages["mask"])
        ex = tiles_close(gp.image(gp.IMG_TILES).cpu().numpy(), op.stages["tiles"], f"frame {f}", shape=(h, w))
        ex = upscale_mask(ex, 1, H, W)
        compare16(helpers.bits16(gp.output(hr.OUTPUT_UPSAMPLE)), up, f"frame {f} upsampled visibility", exclude=ex, outlier_scale=upsample_scale(scale))
    gp.close(); gsc.close()


@pytest.mark.parametrize("name,W,H,scale,spp,params", [
    ("sponza_small", 288, 160, 0, 1, None),
    ("sponza_small", 320, 184, 1, 4, None),
    ("cornell", 230, 150, 0, 2, dict(blur_radius=6, alpha=0.05, ray_length=40.0)),
    ("sponza_small", 171, 121, 0, 3, dict(blur_radius=2)),
    # odd frame size at half resolution: (96 + 0.5) / 193 == 0.5 exactly — a whole column of upsample taps sits on a texel boundary, where
    # an approximate reciprocal picks the other texel (found by tools/fuzz_tolerance.py; the tap addressing keeps the reference's rounding)
    ("sponza_small", 193, 148, 1, 2, None),
    ("cornell", 230, 141, 1, 1, None),
    ("sponza_hard_small", 480, 270, 1, 4, None),
])
def test_ao_tolerance(oracle, hr, ctx, name, W, H, scale, spp, params, n_frames=5):
    import torch
    sd = helpers.scene_data(name)
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    n = n_frames
    frames = helpers.make_frames(oracle, osc, name, W, H, n, 1.5, scale_mips=scale)
    sob, sr, sob_d, sr_d = _tables()
    zbp = synth.z_buffer_params()
    w, h = W >> scale, H >> scale
    kw = dict(params or {})
    gp, op = hr.RayTracedAO(ctx, W, H, scale), oracle.AOPass(w, h, spp=spp, zbp=zbp, **kw)
    for k, v in kw.items():
        setattr(gp.params, k, v)
    gp.params.spp, gp.params.exact = spp, 0
    for f in range(n):
        lvl = (lambda fr: fr["mips"][scale] if scale else fr["gb"])
        cur, prev, full = lvl(frames[f]), lvl(frames[f - 1 if f else 0]), frames[f]["gb"]
        op.render(osc, frames[f]["ubo"], cur, prev, sob, sr, f, full=full if scale else None)
        gp.render(gsc, hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d,
                                       cur_full=helpers.to_cuda(full) if scale else None, z_buffer_params=zbp))
        torch.cuda.synchronize()
        st = op.stages
        mh = (h + 3) // 4
        mask = gp.image(gp.IMG_MASK).cpu().numpy().view(np.uint32)[:spp * mh].reshape(spp, mh, -1)
        assert np.array_equal(mask, st["mask"]), f"frame {f}: AO masks must be bit-exact"
        assert gp.ray_count() == st["rays"]
        ex = tiles_close(gp.image(gp.IMG_TILES).cpu().numpy(), st["tiles"], f"frame {f}", shape=(h, w))
        compare16(helpers.bits16(gp.image(gp.IMG_AO1 if f & 1 else gp.IMG_AO0)), st["temporal"], f"frame {f} temporal AO")
        compare16(helpers.bits16(gp.image(gp.IMG_BLUR1)), st["blur1"], f"frame {f} blurred AO", exclude=ex)
        out = helpers.bits16(gp.output(hr.OUTPUT_UPSAMPLE))
        ref = st["output"] if st["output"].ndim == 2 else st["output"][..., 0]
        # stage-wise (late round 6, as for the a-trous chains): the blur launch(es) and the upsample against the ORACLE's stages run on the very images the GPU launches read
        tiles_np = np.ascontiguousarray(gp.image(gp.IMG_TILES).cpu().numpy().astype(st["tiles"].dtype))
        t_g = np.ascontiguousarray(helpers.bits16(gp.image(gp.IMG_AO1 if f & 1 else gp.IMG_AO0)).reshape(st["temporal"].shape))
        b_g = np.ascontiguousarray(helpers.bits16(gp.image(gp.IMG_BLUR1)).reshape(st["blur1"].shape))
        zb = np.asarray(zbp, np.float32)
        b_ref = oracle.ao_blur(oracle.ao_blur(t_g, cur["depth"], cur["gb2"], tiles_np, zb, (1, 0), op.p["blur_radius"]), cur["depth"], cur["gb2"], tiles_np, zb, (0, 1), op.p["blur_radius"])
        compare16(b_g, b_ref, f"frame {f} blur kernel(s) against the oracle's two blur passes over the same image")
        if scale:
            up_ref = oracle.upsample(full, cur, b_g[..., None], channels=1, sky_value=1.0, power=op.p["power"])
            compare16(out.reshape(ref.shape), up_ref if up_ref.ndim == 2 else up_ref[..., 0], f"frame {f} upsample kernel against the oracle's upsample of the same image")
            ex = upscale_mask(ex, scale, H, W)
        compare16(out, ref, f"frame {f} AO output", exclude=ex, outlier_scale=upsample_scale(scale))
    gp.close(); gsc.close()


@pytest.mark.parametrize("name,W,H,scale,dolly,params", [
    ("sponza_small", 288, 160, 1, 2.0, None),
    ("sponza_small", 224, 128, 0, 1.0, None),
    ("sponza_small", 160, 96, 0, 1.0, dict(approximate_with_ddgi=0, blur_as_input=1, trim=0.5, filter_iterations=3, phi_color=4.0, gi_intensity=1.0, phi_normal=8.0, radius=2)),
])
def test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, name, W, H, scale, dolly, params, n_frames=5):
    """DDGI (exact atlases, tolerance-mode sample) feeding tolerance-mode reflections.  The reflections TRACE (hit shading) has one
    mode and is compared bit for bit while it reads the bit-exact atlases; temporal / a-trous / upsample obey the image rule."""
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections
    from oracle import pyoracle_ddgi as od
    from oracle import pyoracle_reflections as orf
    sd = helpers.scene_data(name)
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
    sky = synth_env.sky_cubemap(16)
    pre, lut = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(16)
    env_np = dict(sky=sky, prefiltered=pre, pre_size=16, pre_levels=5, lut=lut)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(pre), 16, 5, f16(lut))
    n = n_frames
    frames = helpers.make_frames(oracle, osc, name, W, H, n, dolly, scale_mips=scale)
    r01, r003 = np.float16(0.1).view(np.uint16), np.float16(0.03).view(np.uint16)
    for fr in frames:
        for g in [fr["gb"]] + fr.get("mips", [])[1:]:
            ch = g["gb3"][..., 0]
            ch[ch == r01] = r003
    sob, sr, sob_d, sr_d = _tables()
    w, h = W >> scale, H >> scale
    g_ddgi, o_ddgi = api_gi.DDGI(ctx, W, H, ddgi), od.DDGIPass(ddgi)
    g_ddgi.params.exact = 0
    gp = api_reflections.RayTracedReflections(ctx, W, H, scale)
    kw = dict(params or {})
    for k, v in kw.items():
        setattr(gp.params, k, v)
    gp.params.exact = 0
    op = orf.ReflectionsPass(w, h, **kw)
    rng = np.random.RandomState(7)
    for f in range(n):
        lvl = (lambda fr: fr["mips"][scale] if scale else fr["gb"])
        cur, prev, full = lvl(frames[f]), lvl(frames[f - 1 if f else 0]), frames[f]["gb"]
        orient = synth_env.random_orientation(rng)
        cam_delta = (0.0, 0.0, 0.0) if f == 0 else (-dolly, 0.0, 0.0)
        o_ddgi.render(osc, frames[f]["ubo"], full, sky, orient, f)
        irr, dep = o_ddgi.current_read()
        op.render(osc, frames[f]["ubo"], ddgi, cur, prev, sob, sr, f, env_np, irr, dep, camera_delta=cam_delta, full=full if scale else None, ping_pong=bool(f & 1))
        full_d = helpers.to_cuda(full)
        g_ddgi.render(gsc, hr.frame_inputs(full_d, None, frames[f]["ubo"], f, f & 1, sob_d, sr_d), env, orient)
        gp.set_camera_delta(cam_delta)
        gp.render(gsc, hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d, cur_full=full_d), env, g_ddgi)
        torch.cuda.synchronize()
        gi, gd = g_ddgi.current_read()
        assert np.array_equal(helpers.bits16(gi), irr) and np.array_equal(helpers.bits16(gd), dep), f"frame {f}: DDGI atlases are exact in both modes"
        compare16(helpers.bits16(g_ddgi.output()), o_ddgi.stages["output"], f"frame {f} DDGI probe-grid sample", outlier_pixels=DDGI_OUTLIERS)
        st = op.stages
        compare_trace(helpers.bits16(gp.image(gp.IMG_TRACE)), st["trace"], f"frame {f} reflection trace image")
        assert gp.ray_count() == st["rays"]
        ex = tiles_close(gp.image(gp.IMG_TILES).cpu().numpy(), st["tiles"], f"frame {f}", shape=(h, w))
        tc = helpers.bits16(gp.image(gp.IMG_COLOR1 if f & 1 else gp.IMG_COLOR0))
        compare16(tc, st["temporal"], f"frame {f} temporal colour + variance", variance_channels=(3,), outlier_pixels=REFL_OUTLIERS)
        mo = helpers.bits16(gp.image(gp.IMG_MOMENTS1 if f & 1 else gp.IMG_MOMENTS0))
        compare16(mo, st["moments"], f"frame {f} moments (m1, m2, history length, 0)", outlier_pixels=REFL_OUTLIERS)
        at = helpers.bits16(gp.output(hr.OUTPUT_ATROUS))
        # Stage-wise, no allowance at all, ITERATION BY ITERATION: the a-trous launches re-run one at a time (hr_reflections_atrous_iteration — bit-identical to
        # the fused launches of render(): tests/test_gpu_fused.py, and checked on the last image below), each against the ORACLE's iteration on the very image the
        # GPU iteration read.  The reference's luminance weight exp(-|dl| / (phi sqrt(1e-10 + var))) is ill-conditioned where var == 0 (every texel of frame 0):
        # one tolerated fp16 ulp in the output of iteration i re-weights a tap of iteration i + 1 by e^0.6, so a comparison over the whole CHAIN measures how
        # far the reference's own filter spreads a 1-ulp difference (round 6: 4 of 1400 fuzzed sequences had 1-4 texels at 33-137 ulp on frame 0 that way,
        # profiles/r6_e/fuzz_*_6001*.txt), not the kernels; per iteration nothing is left to amplify (tools/refl_outlier_probe.py, docs/EXPERIMENTS.md R5.1 / R5.8 / R6.7).
        gt = gp.image(gp.IMG_TILES).cpu().numpy()
        tiles_np = np.ascontiguousarray(gt.reshape(st["tiles"].shape).astype(st["tiles"].dtype))
        fi_cur = hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d, cur_full=full_d)
        src = tc
        for i in range(op.p["filter_iterations"]):
            gp.atrous_iteration(fi_cur, i)
            torch.cuda.synchronize()
            got_i = helpers.bits16(gp.image(gp.IMG_ATROUS0 if i & 1 else gp.IMG_ATROUS1))
            ref_i = orf.atrous(src, cur, tiles_np, 1 << i, op.p["radius"], op.p["phi_color"], op.p["phi_normal"], op.p["sigma_depth"], op.p["approximate_with_ddgi"])
            compare16(got_i, ref_i, f"frame {f} a-trous iteration {i} against the oracle's iteration on the same input image", variance_channels=(3,), outlier_pixels=0)
            src = got_i
        assert np.array_equal(src, at), f"frame {f}: the iterations one by one must reproduce render()'s a-trous output bit for bit"
        compare16(at, st["atrous"][-1], f"frame {f} a-trous colour + variance", exclude=ex, variance_channels=(3,), outlier_pixels=REFL_OUTLIERS)
        out = helpers.bits16(gp.output(hr.OUTPUT_UPSAMPLE))
        if scale:
            ex = upscale_mask(ex, scale, H, W)
            # stage-wise, no allowance: the oracle's upsample of the GPU's own a-trous output
            compare16(out, oracle.upsample(full, cur, at, channels=4, sky_value=0.0, power=0.0), f"frame {f} upsample kernel against the oracle's upsample of the same image",
                      variance_channels=(3,), outlier_pixels=0)
        compare16(out, st["output"], f"frame {f} reflections output", exclude=ex, variance_channels=(3,), outlier_scale=upsample_scale(scale), outlier_pixels=REFL_OUTLIERS)
    gp.close(); g_ddgi.close(); gsc.close()


# The six fuzzed sequences (of 1784 in round 5, tools/fuzz_tolerance.py) that missed the 99.9 % population bound on a reflections image behind
# the a-trous filter (docs/EXPERIMENTS.md R5.8: 99.83-99.89 %) while the trace kernel's DDGI gathers ran the fast arithmetic; (seed, trial) name
# the draws of helpers.fuzz_configs.  Same runner, same rule, nothing relaxed.
def test_shadows_fuzz_sequence_with_a_spread_ulp(oracle, hr, ctx):
    """tools/fuzz_tolerance.py 6311 #915 (five a-trous iterations of radius 2, phi_normal 8, point light): on frame 4 one texel of the denoised visibility is 51 fp16
    ulp from the oracle although the temporal images agree to 1 ulp and every a-trous launch agrees with the oracle's iteration on its own input — the case the
    shadows chain's counted allowance was stated for (test_shadows_tolerance, ATROUS_OUTLIERS)."""
    c = helpers.fuzz_config(6311, 915)
    assert (c["name"], c["W"], c["H"], c["light"]) == ("sponza_small", 353, 194, "point") and c["shadows"]["filter_iterations"] == 5
    test_shadows_tolerance(oracle, hr, ctx, c["name"], c["W"], c["H"], c["dolly"], c["light"], c["shadows"], n_frames=6)


def test_ao_fuzz_sequence_in_the_dark(oracle, hr, ctx):
    """tools/fuzz_tolerance.py 6301 #279 (quarter-resolution AO, 1 spp, blur radius 2, frame 0): near-black blurred AO (7e-4) inside its 2-ulp bound, raised to
    the power 1.2 by the upsample, left 84 output texels 3-5 fp16 ulp = 2e-7 .. 5e-7 from the oracle — the case OUTPUT_FLOOR was stated for.  The stages
    before the upsample must hold the rule WITHOUT the floor's help being needed: the runner's own comparisons cover them."""
    c = helpers.fuzz_config(6301, 279)
    assert (c["name"], c["W"], c["H"], c["scale"], c["ao_spp"]) == ("sponza_small", 252, 159, 2, 1)
    test_ao_tolerance(oracle, hr, ctx, c["name"], c["W"], c["H"], c["scale"], c["ao_spp"], c["ao"])


def test_long_sequences_do_not_drift(oracle, hr, ctx):
    """The rule holds per frame over 100-frame sequences: the tolerance mode's history (feedback image, moments, history length) is re-read every frame, so an
    error that accumulated through the temporal feedback would show up as a growing share of texels beyond 2 ulp.  (The per-stage comparisons are against the
    oracle's stage images of the SAME frame, each computed from the oracle's own history.)"""
    test_shadows_tolerance(oracle, hr, ctx, "sponza_small", 256, 144, 1.0, "default", None, n_frames=100)
    test_ao_tolerance(oracle, hr, ctx, "sponza_small", 256, 144, 1, 2, None, n_frames=100)
    test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, "sponza_small", 224, 128, 1, 1.0, None, n_frames=100)


FUZZ_SEQUENCES = [(31337, 206), (555, 66), (8088, 21), (8088, 61), (8088, 84), (8088, 159)]


@pytest.mark.parametrize("seed,trial", FUZZ_SEQUENCES)
def test_reflections_fuzz_sequences_that_missed_the_population_bound(oracle, hr, ctx, seed, trial):
    c = helpers.fuzz_config(seed, trial)
    test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, c["name"], c["W"], c["H"], min(c["scale"], 1), c["dolly"], c["reflections"])


def test_ddgi_sample_fuzz_sequence_with_a_pixel_on_a_probe(oracle, hr, ctx):
    """tools/fuzz_tolerance.py 6351 #150 (Cornell room, 254 x 138): on frame 1 pixel (109, 86) shows the corner of the room, which IS a corner probe of the grid fitted
    to the scene's bounds.  The reference's normalize(probe - P) is NaN there and its NaN replacement decides the pixel (net = 0.5); the fast gather's v_max dropped the
    NaN (532 fp16 ulp apart) until ddgi_sample_fast.h learnt to hand such a shading point to the parity gather (case (c) of its redo)."""
    c = helpers.fuzz_config(6351, 150)
    assert (c["name"], c["W"], c["H"], c["scale"]) == ("cornell", 254, 138, 0)
    test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, c["name"], c["W"], c["H"], min(c["scale"], 1), c["dolly"], c["reflections"])


def test_reflections_fuzz_sequence_with_a_flipped_history_tap(oracle, hr, ctx):
    """tools/fuzz_tolerance.py 6361 #385 (half-resolution reflections of a 187 x 212 frame, five a-trous iterations of radius 2, phi_normal 8): on frame 1 ONE pixel of the
    moments image was 92 fp16 ulp from the oracle's — the reflections' temporal kernel decided a history tap's validity with the fast arithmetic — and the a-trous chain
    spread it to 5 pixels beyond the cap (allowance 4) and 0.12 % of the texels beyond 2 ulp.  kf_refl_temporal now re-runs such pixels with the parity verdicts, as the shadows
    and AO temporal kernels do (docs/EXPERIMENTS.md R6.14); the sequence passes under the unchanged rule."""
    c = helpers.fuzz_config(6361, 385)
    assert (c["name"], c["W"], c["H"], c["scale"]) == ("sponza_small", 187, 212, 1)
    test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, c["name"], c["W"], c["H"], min(c["scale"], 1), c["dolly"], c["reflections"])


@pytest.mark.parametrize("tier", ["standard", "hard"])
def test_1080p_bench_frame_tolerance(oracle, hr, ctx, tier):
    """the bench workload (BASELINE configs[1], 1920x1080, 278k triangles) in the mode bench.py times: 3 moving frames; and bench.py's
    hard tier (~2.5 M triangles, grazing sun), where the temporal kernel re-runs pixels with parity verdicts far more often"""
    import torch
    W, H = 1920, 1080
    sd = helpers.scene_data("sponza" if tier == "standard" else "sponza_hard")
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    light = synth.sponza_light() if tier == "standard" else synth.sponza_hard_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(4)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(3)]
    gbs = [gsc.gbuffer(u, W, H) for u in ubos]
    host = [{n: (t.cpu().numpy().view(np.uint16) if t.dtype == torch.float16 else t.cpu().numpy()) for n, t in g.items()} for g in gbs]
    sob, sr, sob_d, sr_d = _tables()
    gp, op = hr.RayTracedShadows(ctx, W, H), oracle.ShadowsPass(W, H)
    gp.params.exact = 0
    for k in range(3):
        op.render(osc, ubos[k], host[k], host[k - 1 if k else 0], sob, sr, k)
        gp.render(gsc, hr.frame_inputs(gbs[k], gbs[k - 1 if k else 0], ubos[k], k, k & 1, sob_d, sr_d))
        torch.cuda.synchronize()
        assert np.array_equal(gp.image(gp.IMG_MASK).cpu().numpy().view(np.uint32), op.stages["mask"]), f"frame {k}: mask"
        ex = tiles_close(gp.image(gp.IMG_TILES).cpu().numpy(), op.stages["tiles"], f"frame {k}", shape=(H, W))
        compare16(helpers.bits16(gp.output(hr.OUTPUT_ATROUS)), op.stages["output"], f"frame {k} denoised visibility + filtered variance", exclude=ex, variance_channels=(1,))
    gp.close(); gsc.close()
