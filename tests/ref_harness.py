"""Host-side sequencing of the REFERENCE'S OWN SHADERS (compiled for the CPU by oracle/pyref.py) — the same dispatch
order the reference's C++ records (ray_traced_shadows.cpp:100-116, ray_traced_ao.cpp:98-112, ...), with numpy arrays in
the oracle's layouts bound as descriptors.  Test infrastructure: used by test_ref_shaders.py and golden/make_ref_golden.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from oracle import pyoracle as oracle
from oracle import pyref

UBO_FIELDS = ("view_inverse", "proj_inverse", "view_proj_inverse", "prev_view_proj", "view_proj", "cam_pos", "current_prev_jitter", "light")
_cache = {}


def shader(rel, defines=()):
    key = (rel, tuple(defines))
    if key not in _cache:
        _cache[key] = pyref.RefShader(rel, defines)
    return _cache[key]


def set_ubo(sh, ubo, block="u_GlobalUBO"):
    for f in UBO_FIELDS:
        if f"{block}.{f}" in sh.regs:
            sh.set(f"{block}.{f}", ubo[f])


def bind_gbuffer(sh, gb, prefix="s_GBuffer", mips=None):
    """gb: dict(gb1, gb2, gb3, depth) numpy arrays; mips: optional list of such dicts (level 0 first)"""
    lv = mips if mips is not None else [gb]
    for name, key, fmt in (("1", "gb1", "rgba8"), ("2", "gb2", "rgba16f"), ("3", "gb3", "rgba16f"), ("Depth", "depth", "r32f")):
        if prefix + name in sh.regs:
            sh.bind(prefix + name, pyref.Tex([m[key] for m in lv], fmt))


def bind_blue_noise(sh, sobol, sr):
    sh.bind("s_SobolSequence", pyref.Tex(np.ascontiguousarray(sobol.reshape(1, 256, 4)), "rgba8"))
    sh.bind("s_ScramblingRankingTile", pyref.Tex(np.ascontiguousarray(sr.reshape(128, 128, 4)), "rgba8"))


def bind_scene(sh, oscene):
    """ray queries answer with the oracle's pinned triangle test (the reference's traversal is the Vulkan driver)"""
    sh.set("u_TopLevelAS", np.uint64(oscene.h.value))
    sh.lib.ref_set_any_hit(C.cast(oracle.lib().orc_any_hit_one, C.c_void_p))


def tile_lists(tile_class):
    ty, tx = np.nonzero(tile_class)
    den = np.ascontiguousarray(np.stack([tx * 8, ty * 8], 1).astype(np.int32))
    sy, sx = np.nonzero(tile_class == 0)
    oth = np.ascontiguousarray(np.stack([sx * 8, sy * 8], 1).astype(np.int32))
    return den, oth


# ------------------------------------------------------------------------------------------------ shadows

def shadows_ray_trace(oscene, ubo, gb, sobol, sr, bias=0.5, num_frames=0):
    sh = shader("shadows/shadows_ray_trace.comp")
    h, w = gb["depth"].shape
    mask = np.zeros(((h + 3) // 4, (w + 7) // 8), np.uint32)
    set_ubo(sh, ubo)
    sh.bind("i_Output", pyref.Tex(mask, "r32ui"))
    bind_gbuffer(sh, gb)
    bind_blue_noise(sh, sobol, sr)
    sh.set_f("u_PushConstants.bias", bias)
    sh.set_u("u_PushConstants.num_frames", num_frames)
    sh.set_i("u_PushConstants.g_buffer_mip", 0)
    bind_scene(sh, oscene)
    sh.dispatch((w + 7) // 8, (h + 3) // 4)
    return mask


def shadows_temporal(ubo, mask, cur, prev, hist_vis_var, hist_moments, alpha=0.01, moments_alpha=0.2):
    sh = shader("shadows/shadows_denoise_reprojection.comp")
    h, w = cur["depth"].shape
    th, tw = (h + 7) // 8, (w + 7) // 8
    out, mom = np.zeros((h, w, 2), np.uint16), np.zeros((h, w, 4), np.uint16)
    den, shd = np.zeros((th * tw, 2), np.int32), np.zeros((th * tw, 2), np.int32)
    set_ubo(sh, ubo)
    sh.bind("i_Output", pyref.Tex(out, "rg16f"))
    sh.bind("i_Moments", pyref.Tex(mom, "rgba16f"))
    bind_gbuffer(sh, cur)
    bind_gbuffer(sh, prev, "s_PrevGBuffer")
    sh.bind("s_Input", pyref.Tex(mask, "r32ui"))
    sh.bind("s_HistoryOutput", pyref.Tex(hist_vis_var, "rg16f"))
    sh.bind("s_HistoryMoments", pyref.Tex(hist_moments, "rgba16f"))
    sh.bind_buffer("DenoiseTileData.coord", den)
    sh.bind_buffer("ShadowTileData.coord", shd)
    for blk in ("DenoiseTileDispatchArgs", "ShadowTileDispatchArgs"):
        sh.set(blk, np.array([0, 1, 1], np.uint32))       # shadows_denoise_reset_args.comp
    sh.set_f("u_PushConstants.alpha", alpha)
    sh.set_f("u_PushConstants.moments_alpha", moments_alpha)
    sh.set_i("u_PushConstants.g_buffer_mip", 0)
    sh.dispatch(tw, th)
    nd = int(np.frombuffer(C.string_at(sh.regs["DenoiseTileDispatchArgs"][0], 12), np.uint32)[0])
    ns = int(np.frombuffer(C.string_at(sh.regs["ShadowTileDispatchArgs"][0], 12), np.uint32)[0])
    return out, mom, den[:nd].copy(), shd[:ns].copy()


def shadows_atrous(inp, gb, den, shd, step, radius=1, phi_visibility=10.0, phi_normal=32.0, sigma_depth=1.0, power=0.0, out=None):
    sh, cp = shader("shadows/shadows_denoise_atrous.comp"), shader("shadows/shadows_denoise_copy_shadow_tiles.comp")
    h, w = inp.shape[:2]
    out = np.zeros((h, w, 2), np.uint16) if out is None else out
    tout = pyref.Tex(out, "rg16f")
    if len(shd):
        cp.bind("i_Output", tout)
        cp.bind_buffer("ShadowTileData.coord", np.ascontiguousarray(shd))
        cp.dispatch(len(shd))
    if len(den):
        sh.bind("i_Output", tout)
        sh.bind("s_Input", pyref.Tex(inp, "rg16f"))
        bind_gbuffer(sh, gb)
        sh.bind_buffer("DenoiseTileData.coord", np.ascontiguousarray(den))
        for k, v in (("radius", radius), ("step_size", step), ("g_buffer_mip", 0)):
            sh.set_i("u_PushConstants." + k, v)
        for k, v in (("phi_visibility", phi_visibility), ("phi_normal", phi_normal), ("sigma_depth", sigma_depth), ("power", power)):
            sh.set_f("u_PushConstants." + k, v)
        sh.dispatch(len(den))
    return out


class RefShadowsPass:
    """RayTracedShadows::render (ray_traced_shadows.cpp:100-116) with the reference's shaders; same state and stage
    dictionary as oracle.ShadowsPass.  Images ping-pong exactly as the reference's do, so texels no dispatch writes keep
    their older content (the oracle restates that as 'unwritten tiles read 0' — compared where it matters)."""

    def __init__(self, w, h, **params):
        self.o = oracle.ShadowsPass(w, h, **params)     # parameter defaults + state layout only; its render() is not used
        self.p = self.o.p
        self.prev_image = np.zeros((h, w, 2), np.uint16)
        self.moments = np.zeros((h, w, 4), np.uint16)
        self.stages = {}

    def render(self, oscene, ubo, cur, prev, sobol, sr, num_frames):
        p = self.p
        mask = shadows_ray_trace(oscene, ubo, cur, sobol, sr, p["bias"], num_frames)
        tv, mom, den, shd = shadows_temporal(ubo, mask, cur, prev, self.prev_image, self.moments, p["alpha"], p["moments_alpha"])
        self.moments = mom
        img, atrous = tv, []
        for i in range(p["filter_iterations"]):
            power = p["power"] if i == p["filter_iterations"] - 1 else 0.0
            img = shadows_atrous(img, cur, den, shd, 1 << i, p["radius"], p["phi_visibility"], p["phi_normal"], p["sigma_depth"], power)
            atrous.append(img)
            if i == p["feedback_iteration"]:
                self.prev_image = img.copy()
        h, w = cur["depth"].shape
        tiles = np.zeros(((h + 7) // 8, (w + 7) // 8), np.uint8)
        tiles[den[:, 1] // 8, den[:, 0] // 8] = 1
        self.stages = dict(mask=mask, temporal=tv, moments=mom, tiles=tiles, denoise_tiles=den, shadow_tiles=shd, atrous=atrous, output=img)
        return img


# ------------------------------------------------------------------------------------------------ ambient occlusion

def ao_ray_trace(oscene, ubo, gb, sobol, sr, bias=0.3, ray_length=7.0, num_frames=0):
    sh = shader("ao/ao_ray_trace.comp")
    h, w = gb["depth"].shape
    mask = np.zeros(((h + 3) // 4, (w + 7) // 8), np.uint32)
    set_ubo(sh, ubo)
    sh.bind("i_Output", pyref.Tex(mask, "r32ui"))
    bind_gbuffer(sh, gb)
    bind_blue_noise(sh, sobol, sr)
    sh.set_u("u_PushConstants.num_frames", num_frames)
    sh.set_f("u_PushConstants.ray_length", ray_length)
    sh.set_f("u_PushConstants.bias", bias)
    sh.set_i("u_PushConstants.g_buffer_mip", 0)
    bind_scene(sh, oscene)
    sh.dispatch((w + 7) // 8, (h + 3) // 4)
    return mask


def ao_temporal(ubo, mask, cur, prev, hist_ao, hist_len, alpha=0.01):
    sh = shader("ao/ao_denoise_reprojection.comp")
    h, w = cur["depth"].shape
    th, tw = (h + 7) // 8, (w + 7) // 8
    out, ln = np.zeros((h, w), np.uint16), np.zeros((h, w), np.uint16)
    den = np.zeros((th * tw, 2), np.int32)
    set_ubo(sh, ubo)
    sh.bind("i_Output", pyref.Tex(out, "r16f"))
    sh.bind("i_HistoryLength", pyref.Tex(ln, "r16f"))
    bind_gbuffer(sh, cur)
    bind_gbuffer(sh, prev, "s_PrevGBuffer")
    sh.bind("s_Input", pyref.Tex(mask, "r32ui"))
    sh.bind("s_PrevAO", pyref.Tex(hist_ao, "r16f"))
    sh.bind("s_PrevHistoryLength", pyref.Tex(hist_len, "r16f"))
    sh.bind_buffer("DenoiseTileData.coord", den)
    sh.set("DenoiseTileDispatchArgs", np.array([0, 1, 1], np.uint32))   # ao_denoise_reset_args.comp
    sh.set_f("u_PushConstants.alpha", alpha)
    sh.set_i("u_PushConstants.g_buffer_mip", 0)
    sh.dispatch(tw, th)
    nd = int(np.frombuffer(C.string_at(sh.regs["DenoiseTileDispatchArgs"][0], 12), np.uint32)[0])
    return out, ln, den[:nd].copy()


def ao_blur(inp, hist_len, gb, den, zbp, direction, radius=4, out=None):
    sh = shader("ao/ao_denoise_bilateral_blur.comp")
    h, w = inp.shape
    out = np.zeros((h, w), np.uint16) if out is None else out
    if len(den):
        sh.bind("i_Output", pyref.Tex(out, "r16f"))
        sh.bind("s_Input", pyref.Tex(inp, "r16f"))
        sh.bind("s_HistoryLength", pyref.Tex(hist_len, "r16f"))
        bind_gbuffer(sh, gb)
        sh.bind_buffer("DenoiseTileData.coord", np.ascontiguousarray(den))
        sh.set("u_PushConstants.z_buffer_params", np.asarray(zbp, np.float32))
        sh.set("u_PushConstants.direction", np.asarray(direction, np.int32))
        sh.set_i("u_PushConstants.radius", radius)
        sh.set_i("u_PushConstants.g_buffer_mip", 0)
        sh.dispatch(len(den))
    return out


def upsample(rel, full_mips, mip_level, lowres, fmt, power=None):
    """ao_upsample.comp / shadows_upsample.comp / reflections_upsample.comp: full_mips = [full-res gb, ..., gb at mip_level]"""
    sh = shader(rel)
    H, W = full_mips[0]["depth"].shape
    ch = {"r16f": 1, "rg16f": 2, "rgba16f": 4}[fmt]
    out = np.zeros((H, W, ch), np.uint16)
    sh.bind("i_Output", pyref.Tex(out, fmt))
    in_fmt = {1: "r16f", 2: "rg16f", 4: "rgba16f"}[1 if lowres.ndim == 2 else lowres.shape[2]]
    sh.bind("s_Input", pyref.Tex(np.ascontiguousarray(lowres), in_fmt))
    bind_gbuffer(sh, full_mips[0], mips=full_mips)
    sh.set_i("u_PushConstants.g_buffer_mip", mip_level)
    if power is not None:
        sh.set_f("u_PushConstants.power", power)
    sh.dispatch((W + 7) // 8, (H + 7) // 8)
    return out


class RefAOPass:
    """RayTracedAO::render (ray_traced_ao.cpp:98-112) with the reference's shaders (1 spp, as the reference traces)"""

    def __init__(self, w, h, zbp, **params):
        self.o = oracle.AOPass(w, h, zbp=zbp, **params)
        self.p, self.zbp = self.o.p, np.asarray(zbp, np.float32)
        self.hist_ao, self.hist_len = np.zeros((h, w), np.uint16), np.zeros((h, w), np.uint16)
        self.stages = {}

    def render(self, oscene, ubo, cur, prev, sobol, sr, num_frames):
        p = self.p
        mask = ao_ray_trace(oscene, ubo, cur, sobol, sr, p["bias"], p["ray_length"], num_frames)
        out, ln, den = ao_temporal(ubo, mask, cur, prev, self.hist_ao, self.hist_len, p["alpha"])
        self.hist_ao, self.hist_len = out, ln
        h, w = cur["depth"].shape
        white = lambda: np.full((h, w), 0x3C00, np.uint16)    # vkCmdClearColorImage(1.0) before each blur (ray_traced_ao.cpp:1048-1055, 1097-1104)
        b0 = ao_blur(out, ln, cur, den, self.zbp, (1, 0), p["blur_radius"], white())
        b1 = ao_blur(b0, ln, cur, den, self.zbp, (0, 1), p["blur_radius"], white())
        tiles = np.zeros(((h + 7) // 8, (w + 7) // 8), np.uint8)
        tiles[den[:, 1] // 8, den[:, 0] // 8] = 1
        self.stages = dict(mask=mask, temporal=out, length=ln, tiles=tiles, blur0=b0, blur1=b1, output=b1)
        return b1


# ------------------------------------------------------------------------------------------------ DDGI (compute stages)

def ddgi_probe_update(ddgi, depth_probe, first_frame, rad, dd, prev_irr, prev_dep):
    """gi_irradiance_probe_update.comp / gi_depth_probe_update.comp (ddgi.cpp:877-899, 935-938): returns the written atlas"""
    sh = shader("gi/gi_depth_probe_update.comp" if depth_probe else "gi/gi_irradiance_probe_update.comp")
    out_i, out_d = np.zeros_like(prev_irr), np.zeros_like(prev_dep)
    sh.bind("i_OutputIrradiance", pyref.Tex(out_i, "rgba16f"))
    sh.bind("i_OutputDepth", pyref.Tex(out_d, "rg16f"))
    sh.bind("s_InputIrradiance", pyref.Tex(prev_irr, "rgba16f", linear=True))
    sh.bind("s_InputDepth", pyref.Tex(prev_dep, "rg16f", linear=True))
    sh.bind("s_InputRadiance", pyref.Tex(rad, "rgba16f"))
    sh.bind("s_InputDirectionDepth", pyref.Tex(dd, "rgba16f"))
    sh.set("ddgi", ddgi.tobytes())
    sh.set_u("u_PushConstants.first_frame", int(first_frame))
    pc = ddgi["probe_counts"]
    sh.dispatch(int(pc[0]) * int(pc[1]), int(pc[2]))
    return out_d if depth_probe else out_i


def ddgi_border_update(ddgi, depth_probe, atlas):
    sh = shader("gi/gi_depth_border_update.comp" if depth_probe else "gi/gi_irradiance_border_update.comp")
    other = np.zeros((4, 4, 2 if not depth_probe else 4), np.uint16)
    sh.bind("i_OutputDepth" if depth_probe else "i_OutputIrradiance", pyref.Tex(atlas, "rg16f" if depth_probe else "rgba16f"))
    sh.bind("i_OutputIrradiance" if depth_probe else "i_OutputDepth", pyref.Tex(other, "rgba16f" if depth_probe else "rg16f"))
    pc = ddgi["probe_counts"]
    sh.dispatch(int(pc[0]) * int(pc[1]), int(pc[2]))
    return atlas


def ddgi_sample_probe_grid(ubo, ddgi, gb, gi_intensity, irr, dep):
    sh = shader("gi/gi_sample_probe_grid.comp")
    h, w = gb["depth"].shape
    out = np.zeros((h, w, 4), np.uint16)
    set_ubo(sh, ubo)
    sh.bind("i_Output", pyref.Tex(out, "rgba16f"))
    sh.bind("s_Irradiance", pyref.Tex(irr, "rgba16f", linear=True))
    sh.bind("s_Depth", pyref.Tex(dep, "rg16f", linear=True))
    bind_gbuffer(sh, gb)
    sh.set("ddgi", ddgi.tobytes())
    sh.set_i("u_PushConstants.g_buffer_mip", 0)
    sh.set_f("u_PushConstants.gi_intensity", gi_intensity)
    sh.dispatch((w + 7) // 8, (h + 7) // 8)
    return out


# ------------------------------------------------------------------------------------------------ reflections (denoiser)

def reflections_temporal(ubo, inp, cur, prev, hist_color, hist_moments, camera_delta, alpha, moments_alpha, approx):
    sh = shader("reflections/reflections_denoise_reprojection.comp")
    h, w = cur["depth"].shape
    th, tw = (h + 7) // 8, (w + 7) // 8
    oc, om = np.zeros((h, w, 4), np.uint16), np.zeros((h, w, 4), np.uint16)
    den, cpy = np.zeros((th * tw, 2), np.int32), np.zeros((th * tw, 2), np.int32)
    set_ubo(sh, ubo)
    sh.bind("i_Output", pyref.Tex(oc, "rgba16f"))
    sh.bind("i_Moments", pyref.Tex(om, "rgba16f"))
    bind_gbuffer(sh, cur)
    bind_gbuffer(sh, prev, "s_PrevGBuffer")
    sh.bind("s_Input", pyref.Tex(inp, "rgba16f"))
    sh.bind("s_HistoryOutput", pyref.Tex(hist_color, "rgba16f"))
    sh.bind("s_HistoryMoments", pyref.Tex(hist_moments, "rgba16f"))
    sh.bind_buffer("DenoiseTileData.coord", den)
    sh.bind_buffer("CopyTileData.coord", cpy)
    for blk in ("DenoiseTileDispatchArgs", "CopyTileDispatchArgs"):
        sh.set(blk, np.array([0, 1, 1], np.uint32))       # reflections_denoise_reset_args.comp
    sh.set("u_PushConstants.camera_delta", np.asarray(camera_delta, np.float32))
    sh.set_f("u_PushConstants.frame_time", 0.0)
    sh.set_f("u_PushConstants.alpha", alpha)
    sh.set_f("u_PushConstants.moments_alpha", moments_alpha)
    sh.set_i("u_PushConstants.g_buffer_mip", 0)
    sh.set_i("u_PushConstants.approximate_with_ddgi", int(approx))
    sh.dispatch(tw, th)
    nd = int(np.frombuffer(C.string_at(sh.regs["DenoiseTileDispatchArgs"][0], 12), np.uint32)[0])
    nc = int(np.frombuffer(C.string_at(sh.regs["CopyTileDispatchArgs"][0], 12), np.uint32)[0])
    return oc, om, den[:nd].copy(), cpy[:nc].copy()


def reflections_atrous(inp, gb, den, cpy, step, radius, phi_color, phi_normal, sigma_depth, approx):
    sh, cp = shader("reflections/reflections_denoise_atrous.comp"), shader("reflections/reflections_denoise_copy_tiles.comp")
    h, w = inp.shape[:2]
    out = np.zeros((h, w, 4), np.uint16)
    tout, tin = pyref.Tex(out, "rgba16f"), pyref.Tex(inp, "rgba16f")
    if len(cpy):
        cp.bind("i_Output", tout)
        cp.bind("s_Input", tin)
        cp.bind_buffer("CopyTileData.coord", np.ascontiguousarray(cpy))
        cp.dispatch(len(cpy))
    if len(den):
        sh.bind("i_Output", tout)
        sh.bind("s_Input", tin)
        bind_gbuffer(sh, gb)
        sh.bind_buffer("DenoiseTileData.coord", np.ascontiguousarray(den))
        for k, v in (("radius", radius), ("step_size", step), ("g_buffer_mip", 0), ("approximate_with_ddgi", int(approx))):
            sh.set_i("u_PushConstants." + k, v)
        for k, v in (("phi_color", phi_color), ("phi_normal", phi_normal), ("sigma_depth", sigma_depth)):
            sh.set_f("u_PushConstants." + k, v)
        sh.dispatch(len(den))
    return out


# ------------------------------------------------------------------------------------------------ TAA

def taa_resolve(color, prev, gb, jitter, feedback_min=0.88, feedback_max=0.97, sharpen=True):
    """taa.comp (temporal_aa.cpp:118-150): s_Current / s_Prev bilinear (:255), velocity / depth nearest"""
    sh = shader("taa.comp")
    h, w = color.shape[:2]
    out = np.zeros((h, w, 4), np.uint16)
    sh.bind("i_Color", pyref.Tex(out, "rgba16f"))
    sh.bind("s_Current", pyref.Tex(color, "rgba16f", linear=True))
    sh.bind("s_Prev", pyref.Tex(prev, "rgba16f", linear=True))
    sh.bind("s_Velocity", pyref.Tex(gb["gb2"], "rgba16f"))
    sh.bind("s_Depth", pyref.Tex(gb["depth"], "r32f"))
    sh.set("u_TexelSize", np.array([np.float32(1.0) / np.float32(w), np.float32(1.0) / np.float32(h), w, h], np.float32))
    sh.set("u_CurrentPrevJitter", np.asarray(jitter, np.float32))
    sh.set("u_TimeParams", np.zeros(4, np.float32))
    sh.set_f("u_FeedbackMin", feedback_min)
    sh.set_f("u_FeedbackMax", feedback_max)
    sh.set_i("u_Sharpen", int(sharpen))
    sh.dispatch((w + 31) // 32, (h + 31) // 32)
    return out
