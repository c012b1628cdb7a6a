"""ORACLE — TEST INFRASTRUCTURE ONLY.

Host-side sequencing of the REFERENCE'S OWN SHADERS (compiled for the CPU by oracle/pyref.py) — the same dispatch
order the reference's C++ records (ray_traced_shadows.cpp:100-116, ray_traced_ao.cpp:98-112, ...), with numpy arrays in
the oracle's layouts bound as descriptors.  Used by tests/test_ref_shaders.py, tests/golden/make_ref_golden.py and the
cpu_baseline leg of bench.py ("reference_shaders").
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import pyoracle as oracle
from . import pyref

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


# ------------------------------------------------------------------------------------------------ ray-tracing pipelines

_pipes = {}


def pipeline(name, stages, variant=None):
    if name not in _pipes:
        _pipes[name] = pyref.RefPipeline(name, stages, variant=variant)
    return _pipes[name]


class RefScene:
    """The reference's scene descriptor set (scene_descriptor_set.glsl:58-91) built from a synth.SceneData: one instance
    with an identity model matrix, one mesh, one BLAS geometry per triangle (SubmeshInfo = (triangle, material)), no
    textures (all texture indices -1).  Layouts are the shim's C++ structs (members in declaration order, no padding)."""

    def __init__(self, sd):
        n = sd.n_tris
        v = np.zeros((n * 3, 5, 4), np.float32)               # Vertex: position, tex_coord, normal, tangent, bitangent
        v[:, 0, :3], v[:, 0, 3] = sd.verts.reshape(-1, 3), 1.0
        v[:, 2, :3] = sd.normals.reshape(-1, 3)
        v[:, 3, 0], v[:, 4, 1] = 1.0, 1.0
        self.vertices = np.ascontiguousarray(v)
        self.indices = np.arange(n * 3, dtype=np.uint32)
        self.submesh = np.ascontiguousarray(np.stack([np.arange(n, dtype=np.uint32), sd.tri_material.astype(np.uint32)], 1))
        m = np.zeros((len(sd.materials), 20), np.float32)     # Material: 2 x ivec4, albedo, emissive, roughness_metallic
        mi = m.view(np.int32)
        mi[:, 0:8] = -1
        m[:, 8:11], m[:, 11] = sd.materials[:, 0:3], 1.0
        m[:, 12:15] = sd.materials[:, 5:8]
        m[:, 16], m[:, 17] = sd.materials[:, 4], sd.materials[:, 3]
        self.textures = []
        if getattr(sd, "material_textures", None) is not None:
            if sd.uvs is not None:
                v[:, 1, :2] = sd.uvs.reshape(-1, 2)
            if sd.tangents is not None:
                v[:, 3, :3] = sd.tangents.reshape(-1, 3)
                v[:, 4, :3] = sd.tangents.reshape(-1, 3)
            self.vertices = np.ascontiguousarray(v)
            mtx = np.asarray(sd.material_textures, np.int32)
            mi[:, 0:4] = mtx[:, 0:4]            # texture_indices0: albedo, normals, roughness, metallic
            mi[:, 6], mi[:, 7] = mtx[:, 4], mtx[:, 5]   # texture_indices1.z / .w: roughness / metallic channel
            # pinned sampler of s_Textures[]: bilinear, repeat (see orc_shading.h sample_texture)
            self.textures = [pyref.Tex(np.ascontiguousarray(t, np.uint8), "rgba8", linear=True, repeat=True) for t in sd.textures]
            self.texture_table = np.array([t.ptr for t in self.textures], np.uint64)
        self.materials = np.ascontiguousarray(m)
        inst = np.zeros(17, np.float32)
        inst[[0, 5, 10, 15]] = 1.0
        self.instances = np.ascontiguousarray(inst)            # mat4 model + uint mesh_idx (= 0)

    def bind(self, pipe, oscene):
        pipe.set_all("Materials.data", np.uint64(self.materials.ctypes.data))
        pipe.set_all("Instances.data", np.uint64(self.instances.ctypes.data))
        pipe.set_at_all("Vertices", np.uint64(self.vertices.ctypes.data))
        pipe.set_at_all("Indices", np.uint64(self.indices.ctypes.data))
        pipe.set_at_all("SubmeshInfo", np.uint64(self.submesh.ctypes.data))
        if self.textures:
            pipe.set_all("s_Textures", np.uint64(self.texture_table.ctypes.data))
        pipe.set_all("u_TopLevelAS", np.uint64(oscene.h.value))
        pipe.lib.ref_set_any_hit(C.cast(oracle.lib().orc_any_hit_one, C.c_void_p))
        pipe.lib.ref_set_closest_hit(C.cast(oracle.lib().orc_closest_hit_one, C.c_void_p))
        if hasattr(pipe.lib, "ref_set_instance_map"):
            pipe.lib.ref_set_instance_map(None, None)     # one instance
        pipe._keep["scene"] = self


class RefInstancedScene:
    """The reference's scene descriptor set for a synth.InstancedSceneData: Instances.data[i] = { model_matrix, mesh_idx }, one vertex / index /
    submesh buffer per MESH (object space), one BLAS geometry per mesh triangle — fetch_hit_info / fetch_triangle / transform_vertex
    (scene_descriptor_set.glsl:102-160) then run as the reference wrote them.  The ray queries are answered by the oracle's BVH over the
    flattened world-space vertices; the shim maps the hit triangle to (instance, geometry) (refshim/runtime_rt.inc)."""

    def __init__(self, isd, matrices=None):
        mats = isd.matrices() if matrices is None else np.ascontiguousarray(np.asarray(matrices, np.float32).reshape(-1, 16))
        self.vertices, self.indices, self.submesh = [], [], []
        for me in isd.meshes:
            n = me.n_tris
            v = np.zeros((n * 3, 5, 4), np.float32)
            v[:, 0, :3], v[:, 0, 3] = np.asarray(me.verts, np.float32).reshape(-1, 3), 1.0
            v[:, 2, :3] = np.asarray(me.normals, np.float32).reshape(-1, 3)
            v[:, 3, 0], v[:, 4, 1] = 1.0, 1.0
            if me.uvs is not None:
                v[:, 1, :2] = np.asarray(me.uvs, np.float32).reshape(-1, 2)
            if me.tangents is not None:
                v[:, 3, :3] = np.asarray(me.tangents, np.float32).reshape(-1, 3)
                v[:, 4, :3] = np.asarray(me.tangents, np.float32).reshape(-1, 3)
            self.vertices.append(np.ascontiguousarray(v))
            self.indices.append(np.arange(n * 3, dtype=np.uint32))
            self.submesh.append(np.ascontiguousarray(np.stack([np.arange(n, dtype=np.uint32), np.asarray(me.tri_material, np.uint32)], 1)))
        mt = np.asarray(isd.materials, np.float32)
        m = np.zeros((len(mt), 20), np.float32)
        mi = m.view(np.int32)
        mi[:, 0:8] = -1
        m[:, 8:11], m[:, 11] = mt[:, 0:3], 1.0
        m[:, 12:15] = mt[:, 5:8]
        m[:, 16], m[:, 17] = mt[:, 4], mt[:, 3]
        self.textures = []
        if isd.material_textures is not None:
            mtx = np.asarray(isd.material_textures, np.int32)
            mi[:, 0:4] = mtx[:, 0:4]
            mi[:, 6], mi[:, 7] = mtx[:, 4], mtx[:, 5]
            self.textures = [pyref.Tex(np.ascontiguousarray(t, np.uint8), "rgba8", linear=True, repeat=True) for t in isd.textures]
            self.texture_table = np.array([t.ptr for t in self.textures], np.uint64)
        self.materials = np.ascontiguousarray(m)
        inst = np.zeros((len(isd.instances), 17), np.float32)
        inst[:, :16] = mats
        inst.view(np.uint32)[:, 16] = [k for _, k, _ in isd.instances]
        self.instances = np.ascontiguousarray(inst)
        first, _, _, n = isd.layout()
        self.first_tri = np.ascontiguousarray(first, np.uint32)
        self.tri_instance = np.ascontiguousarray(np.repeat(np.arange(len(n), dtype=np.uint32), n))

    def bind(self, pipe, oscene):
        pipe.set_all("Materials.data", np.uint64(self.materials.ctypes.data))
        pipe.set_all("Instances.data", np.uint64(self.instances.ctypes.data))
        for k in range(len(self.vertices)):
            pipe.set_at_all("Vertices", np.uint64(self.vertices[k].ctypes.data), 8 * k)
            pipe.set_at_all("Indices", np.uint64(self.indices[k].ctypes.data), 8 * k)
            pipe.set_at_all("SubmeshInfo", np.uint64(self.submesh[k].ctypes.data), 8 * k)
        if self.textures:
            pipe.set_all("s_Textures", np.uint64(self.texture_table.ctypes.data))
        pipe.set_all("u_TopLevelAS", np.uint64(oscene.h.value))
        pipe.lib.ref_set_any_hit(C.cast(oracle.lib().orc_any_hit_one, C.c_void_p))
        pipe.lib.ref_set_closest_hit(C.cast(oracle.lib().orc_closest_hit_one, C.c_void_p))
        if hasattr(pipe.lib, "ref_set_instance_map"):   # ray-tracing pipelines; a ray QUERY (shadows / AO) never asks which instance it hit
            pipe.lib.ref_set_instance_map(C.c_void_p(self.tri_instance.ctypes.data), C.c_void_p(self.first_tri.ctypes.data))
        pipe._keep["scene"] = self


def cube_tex(sky):
    """[6][S][S][4] fp16 -> a cube texture of the shim (faces stacked along y)"""
    S = sky.shape[1]
    return pyref.Tex(np.ascontiguousarray(sky.reshape(6 * S, S, 4)), "rgba16f", layers=6)


def mat4_from_3x3(cm9):
    m = np.zeros((4, 4), np.float32)
    m[:3, :3] = np.asarray(cm9, np.float32).reshape(3, 3)     # column-major in, column-major out
    m[3, 3] = 1.0
    return np.ascontiguousarray(m.reshape(16))


def ddgi_ray_trace(oscene, rscene, ubo, ddgi, orientation, num_frames, infinite_bounces, gi_intensity, sky, prev_irr, prev_dep):
    """gi_ray_trace.rgen / .rchit / .rmiss (ddgi.cpp:788-812): returns (radiance, direction_distance) [probes][rays][4] fp16"""
    pipe = pipeline("gi_ray_trace", [("gi/gi_ray_trace.rgen", 0), ("gi/gi_ray_trace.rchit", 1), ("gi/gi_ray_trace.rmiss", 2)])
    n_probes, R = int(np.prod(ddgi["probe_counts"])), int(ddgi["rays_per_probe"])
    rad, dd = np.zeros((n_probes, R, 4), np.uint16), np.zeros((n_probes, R, 4), np.uint16)
    rscene.bind(pipe, oscene)
    for f in UBO_FIELDS:
        pipe.set_all("ubo." + f, ubo[f])
    pipe.set_all("ddgi", ddgi.tobytes())
    pipe.bind_all("i_Radiance", pyref.Tex(rad, "rgba16f"))
    pipe.bind_all("i_DirectionDistance", pyref.Tex(dd, "rgba16f"))
    pipe.bind_all("s_Cubemap", cube_tex(sky))
    pipe.bind_all("s_Irradiance", pyref.Tex(prev_irr, "rgba16f", linear=True))
    pipe.bind_all("s_Depth", pyref.Tex(prev_dep, "rg16f", linear=True))
    pipe.set_all("u_PushConstants.random_orientation", mat4_from_3x3(orientation))
    pipe.set_all("u_PushConstants.num_frames", np.uint32(num_frames))
    pipe.set_all("u_PushConstants.infinite_bounces", np.uint32(int(infinite_bounces)))
    pipe.set_all("u_PushConstants.gi_intensity", np.float32(gi_intensity))
    pipe.trace_rays(R, n_probes)
    return rad, dd


def prefiltered_tex(pre, size, levels):
    """concatenated [6][s][s][4] levels (s = size >> level) -> a cube texture with a mip chain"""
    lv, off = [], 0
    flat = pre.reshape(-1)
    for l in range(levels):
        s = size >> l
        n = 6 * s * s * 4
        lv.append(np.ascontiguousarray(flat[off:off + n].reshape(6 * s, s, 4)))
        off += n
    return pyref.Tex(lv, "rgba16f", layers=6)


def reflections_ray_trace(oscene, rscene, ubo, ddgi, cur, sobol, sr, prm, env, irr, dep):
    """reflections_ray_trace.rgen / .rchit / .rmiss (ray_traced_reflections.cpp ray_trace()): prm = oracle TraceParams"""
    pipe = pipeline("reflections_ray_trace", [("reflections/reflections_ray_trace.rgen", 0), ("reflections/reflections_ray_trace.rchit", 1),
                                              ("reflections/reflections_ray_trace.rmiss", 2)])
    h, w = cur["depth"].shape
    out = np.zeros((h, w, 4), np.uint16)
    rscene.bind(pipe, oscene)
    for f in UBO_FIELDS:
        pipe.set_all("ubo." + f, ubo[f])
        pipe.set_all("u_GlobalUBO." + f, ubo[f])
    pipe.set_all("ddgi", ddgi.tobytes())
    pipe.bind_all("i_Color", pyref.Tex(out, "rgba16f"))
    for name, key, fmt in (("1", "gb1", "rgba8"), ("2", "gb2", "rgba16f"), ("3", "gb3", "rgba16f"), ("Depth", "depth", "r32f")):
        pipe.bind_all("s_GBuffer" + name, pyref.Tex(cur[key], fmt))
    pipe.bind_all("s_SobolSequence", pyref.Tex(np.ascontiguousarray(sobol.reshape(1, 256, 4)), "rgba8"))
    pipe.bind_all("s_ScramblingRankingTile", pyref.Tex(np.ascontiguousarray(sr.reshape(128, 128, 4)), "rgba8"))
    pipe.bind_all("s_Cubemap", cube_tex(env["sky"]))
    pipe.bind_all("s_Prefiltered", prefiltered_tex(env["prefiltered"], env["pre_size"], env["pre_levels"]))
    pipe.bind_all("s_BRDF", pyref.Tex(env["lut"], "rg16f"))
    pipe.bind_all("s_Irradiance", pyref.Tex(irr, "rgba16f", linear=True))
    pipe.bind_all("s_Depth", pyref.Tex(dep, "rg16f", linear=True))
    for k, v in (("bias", np.float32(prm.bias)), ("trim", np.float32(prm.trim)), ("num_frames", np.uint32(prm.num_frames)), ("g_buffer_mip", np.int32(0)),
                 ("sample_gi", np.int32(prm.sample_gi)), ("approximate_with_ddgi", np.int32(prm.approximate_with_ddgi)),
                 ("gi_intensity", np.float32(prm.gi_intensity)), ("rough_ddgi_intensity", np.float32(prm.rough_ddgi_intensity)),
                 ("ibl_indirect_specular_intensity", np.float32(prm.ibl_indirect_specular_intensity))):
        pipe.set_all("u_PushConstants." + k, v)
    pipe.trace_rays(w, h)
    return out


def ground_truth(oscene, rscene, ubo, sky, w, h, num_frames, prev, roughness_multiplier=1.0, max_ray_bounces=2, trace_indirect=False):
    """ground_truth_path_trace.rgen / .rchit / .rmiss (ground_truth_path_tracer.cpp:44-111); prev / result [h][w][4] fp16.
    trace_indirect: the variant with the recursive traceRayEXT of rchit:95-105 un-commented (translate.VARIANTS["bounces"])"""
    stages = [("ground_truth/ground_truth_path_trace.rgen", 0), ("ground_truth/ground_truth_path_trace.rchit", 1), ("ground_truth/ground_truth_path_trace.rmiss", 2)]
    pipe = pipeline("ground_truth_path_trace_bounces", stages, variant="bounces") if trace_indirect else pipeline("ground_truth_path_trace", stages)
    out = np.zeros((h, w, 4), np.uint16)
    rscene.bind(pipe, oscene)
    for f in UBO_FIELDS:
        pipe.set_all("ubo." + f, ubo[f])
    pipe.bind_all("i_CurrentColor", pyref.Tex(out, "rgba16f"))
    pipe.bind_all("i_PreviousColor", pyref.Tex(prev, "rgba16f"))
    pipe.bind_all("s_Cubemap", cube_tex(sky))
    pipe.set_all("u_PushConstants.num_frames", np.uint32(num_frames))
    pipe.set_all("u_PushConstants.max_ray_bounces", np.uint32(max_ray_bounces))
    pipe.set_all("u_PushConstants.roughness_multiplier", np.float32(roughness_multiplier))
    pipe.trace_rays(w, h)
    return out


# ------------------------------------------------------------------------------------------------ deferred composite

def deferred_shade(ubo, gb, shadow, ao, reflections, gi, flags, sh9, env):
    """deferred.frag as a full-screen pass (deferred_shading.cpp); flags: bit0 shadow, 1 ao, 2 reflections, 3 gi.
    Returns RGBA16F bit patterns (the colour attachment is RGBA16F: fp32 -> fp16 round to nearest even)."""
    sh = shader("deferred.frag")
    h, w = gb["depth"].shape
    set_ubo(sh, ubo)
    bind_gbuffer(sh, gb)
    as_tex = lambda a, fmts: pyref.Tex(np.ascontiguousarray(a), fmts[1 if a.ndim == 2 else a.shape[2]])
    f1 = {1: "r16f", 2: "rg16f", 4: "rgba16f"}
    zero = np.zeros((h, w, 4), np.uint16)
    sh.bind("s_Shadow", as_tex(shadow if shadow is not None else zero, f1))
    sh.bind("s_AO", as_tex(ao if ao is not None else zero, f1))
    sh.bind("s_Reflections", as_tex(reflections if reflections is not None else zero, f1))
    sh.bind("s_GI", as_tex(gi if gi is not None else zero, f1))
    sh.bind("s_IrradianceSH", pyref.Tex(np.ascontiguousarray(np.asarray(sh9, np.float32).reshape(1, 9, 4)), "rgba32f"))
    sh.bind("s_Prefiltered", prefiltered_tex(env["prefiltered"], env["pre_size"], env["pre_levels"]))
    sh.bind("s_BRDF", pyref.Tex(env["lut"], "rg16f"))
    for i, k in enumerate(("shadow", "ao", "reflections", "gi")):
        sh.set_i("u_PushConstants." + k, (flags >> i) & 1)
    out = sh.fragments(w, h, "FS_IN_TexCoord", "FS_OUT_Color")
    return np.ascontiguousarray(out.astype(np.float16)).view(np.uint16)


# ------------------------------------------------------------------------------------------------ whole passes on the reference shaders

class RefDDGIPass:
    """DDGI::render (ddgi.cpp:89-104): ray trace -> probe updates -> border updates -> sample grid, reference shaders only"""

    def __init__(self, ddgi, sd, **params):
        from oracle import pyoracle_ddgi as od
        self.o = od.DDGIPass(ddgi, **params)          # parameter defaults + atlas allocation only
        self.ddgi, self.p = ddgi, self.o.p
        self.irr, self.dep = self.o.irr, self.o.dep
        self.first_frame, self.ping_pong = True, False
        self.rscene = RefScene(sd)
        self.stages = {}

    def current_read(self):
        i = int(not self.ping_pong)
        return self.irr[i], self.dep[i]

    def render(self, oscene, ubo, cur, sky, orientation, num_frames):
        p, d = self.p, self.ddgi
        rd, wr = int(not self.ping_pong), int(self.ping_pong)
        inf = p["infinite_bounces"] and not self.first_frame
        rad, dd = ddgi_ray_trace(oscene, self.rscene, ubo, d, orientation, num_frames, inf, p["infinite_bounce_intensity"], sky, self.irr[rd], self.dep[rd])
        self.irr[wr] = ddgi_border_update(d, False, ddgi_probe_update(d, False, self.first_frame, rad, dd, self.irr[rd], self.dep[rd]))
        self.dep[wr] = ddgi_border_update(d, True, ddgi_probe_update(d, True, self.first_frame, rad, dd, self.irr[rd], self.dep[rd]))
        out = ddgi_sample_probe_grid(ubo, d, cur, p["gi_intensity"], self.irr[wr], self.dep[wr])
        self.stages = dict(radiance=rad, direction_distance=dd, irradiance=self.irr[wr], depth=self.dep[wr], output=out)
        self.first_frame = False
        self.ping_pong = not self.ping_pong
        return out


class RefReflectionsPass:
    """RayTracedReflections::render (ray_traced_reflections.cpp:107-123), reference shaders only"""

    def __init__(self, w, h, sd, **params):
        from oracle import pyoracle_reflections as orf
        self.orf = orf
        self.o = orf.ReflectionsPass(w, h, **params)   # parameter defaults + image allocation only
        self.p = self.o.p
        self.color, self.moments, self.prev_image = self.o.color, self.o.moments, self.o.prev_image
        self.ping_pong = False
        self.rscene = RefScene(sd)
        self.stages = {}

    def render(self, oscene, ubo, ddgi, cur, prev, sobol, sr, num_frames, env, irr, dep, camera_delta=(0, 0, 0), full_mips=None):
        p = self.p
        pp = int(self.ping_pong)
        tp = self.orf.TraceParams(p["bias"], p["trim"], num_frames, int(p["sample_gi"]), int(p["approximate_with_ddgi"]), p["gi_intensity"],
                                  p["rough_ddgi_intensity"], p["ibl_indirect_specular_intensity"])
        traced = reflections_ray_trace(oscene, self.rscene, ubo, ddgi, cur, sobol, sr, tp, env, irr, dep)
        hist = self.prev_image if p["blur_as_input"] else self.color[1 - pp]
        oc, om, den, cpy = reflections_temporal(ubo, traced, cur, prev, hist, self.moments[1 - pp], camera_delta, p["alpha"], p["moments_alpha"],
                                                p["approximate_with_ddgi"])
        self.color[pp], self.moments[pp] = oc, om
        img, its = oc, []
        for i in range(p["filter_iterations"]):
            img = reflections_atrous(img, cur, den, cpy, 1 << i, p["radius"], p["phi_color"], p["phi_normal"], p["sigma_depth"], p["approximate_with_ddgi"])
            its.append(img)
            if i == p["feedback_iteration"] and p["blur_as_input"]:
                self.prev_image = img.copy()
        up = upsample("reflections/reflections_upsample.comp", full_mips, len(full_mips) - 1, img, "rgba16f") if full_mips else None
        h, w = cur["depth"].shape
        tiles = np.zeros(((h + 7) // 8, (w + 7) // 8), np.uint8)
        tiles[den[:, 1] // 8, den[:, 0] // 8] = 1
        self.stages = dict(trace=traced, temporal=oc, moments=om, tiles=tiles, atrous=its, upsample=up, output=img if up is None else up)
        self.ping_pong = not self.ping_pong
        return self.stages["output"]


def tone_map(color, single_channel=False, exposure=1.0):
    """tone_map.frag as a full-screen pass; s_Color is TemporalAA's output descriptor (bilinear, temporal_aa.cpp:255)"""
    sh = shader("tone_map.frag")
    h, w = color.shape[:2]
    sh.bind("s_Color", pyref.Tex(color, "rgba16f", linear=True))
    sh.set_i("u_PushConstants.single_channel", int(single_channel))
    sh.set_f("u_PushConstants.exposure", exposure)
    return sh.fragments(w, h, "inUV", "FS_OUT_Color")
