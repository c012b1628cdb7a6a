"""Synthetic environment / DDGI inputs: the sky cubemap that stands in for the reference's
Hosek-Wilkie sky render (common.h:91-103, absent framework), the DDGI uniform block
(ddgi.cpp:14-32, :150-169, :197-201, :738-763) and the per-frame probe-ray rotation (ddgi.cpp:788)."""
from __future__ import annotations

import math

import numpy as np

DDGI_DTYPE = np.dtype([
    ("grid_start_position", np.float32, 3), ("grid_step", np.float32, 3), ("probe_counts", np.int32, 3),
    ("max_distance", np.float32), ("depth_sharpness", np.float32), ("hysteresis", np.float32), ("normal_bias", np.float32),
    ("energy_preservation", np.float32), ("irradiance_probe_side_length", np.int32), ("irradiance_texture_width", np.int32),
    ("irradiance_texture_height", np.int32), ("depth_probe_side_length", np.int32), ("depth_texture_width", np.int32),
    ("depth_texture_height", np.int32), ("rays_per_probe", np.int32), ("visibility_test", np.int32),
])
assert DDGI_DTYPE.itemsize == 88


def ddgi_uniforms(bounds_lo, bounds_hi, probe_distance=None, probe_counts=None, rays_per_probe=256, normal_bias=0.25,
                  hysteresis=0.98, depth_sharpness=50.0, energy_preservation=0.85, visibility_test=True,
                  irradiance_oct_size=8, depth_oct_size=16) -> np.ndarray:
    """DDGI::initialize_probe_grid (ddgi.cpp:150-169) + update_properties_ubo (:738-763).

    Either ``probe_distance`` (reference behaviour: counts = ivec3(extent / distance) + 2) or explicit
    ``probe_counts`` (BASELINE.json configs[4]: 16x8x16; the step is then extent / (counts - 2))."""
    lo, hi = np.asarray(bounds_lo, np.float32), np.asarray(bounds_hi, np.float32)
    ext = hi - lo
    if probe_counts is None:
        counts = (ext / np.float32(probe_distance)).astype(np.int32) + 2
        step = np.full(3, probe_distance, np.float32)
        max_distance = np.float32(probe_distance) * np.float32(1.5)
    else:
        counts = np.asarray(probe_counts, np.int32)
        step = (ext / np.maximum(counts - 2, 1)).astype(np.float32)
        max_distance = np.float32(step.max()) * np.float32(1.5)
    u = np.zeros((), DDGI_DTYPE)
    u["grid_start_position"], u["grid_step"], u["probe_counts"] = lo, step, counts
    u["max_distance"], u["depth_sharpness"], u["hysteresis"] = max_distance, depth_sharpness, hysteresis
    u["normal_bias"], u["energy_preservation"] = normal_bias, energy_preservation
    u["irradiance_probe_side_length"], u["depth_probe_side_length"] = irradiance_oct_size, depth_oct_size
    u["irradiance_texture_width"] = (irradiance_oct_size + 2) * counts[0] * counts[1] + 2     # ddgi.cpp:197-201
    u["irradiance_texture_height"] = (irradiance_oct_size + 2) * counts[2] + 2
    u["depth_texture_width"] = (depth_oct_size + 2) * counts[0] * counts[1] + 2
    u["depth_texture_height"] = (depth_oct_size + 2) * counts[2] + 2
    u["rays_per_probe"], u["visibility_test"] = rays_per_probe, int(visibility_test)
    return u


def random_orientation(rng: np.random.RandomState) -> np.ndarray:
    """glm::mat4_cast(glm::angleAxis(U(0,1) * 2pi, normalize(U(-1,1)^3))) -> column-major 3x3 (ddgi.cpp:788)."""
    angle = rng.uniform(0.0, 1.0) * 2.0 * math.pi
    axis = rng.uniform(-1.0, 1.0, 3)
    axis /= np.linalg.norm(axis)
    x, y, z = axis
    c, s = math.cos(angle), math.sin(angle)
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    return np.ascontiguousarray(R.T.reshape(9), np.float32)  # column-major


def sky_cubemap(size: int = 32, sun_dir=(0.3, 0.9, 0.2), intensity: float = 1.0) -> np.ndarray:
    """[6][S][S][4] fp16 (as uint16 bit patterns): blue gradient + warm lobe around the sun + dark ground.
    Face order / orientation follows the Vulkan cube-map selection rule (+X -X +Y -Y +Z -Z)."""
    S = size
    t = (np.arange(S, dtype=np.float64) + 0.5) / S * 2 - 1
    sc, tc = np.meshgrid(t, t)  # sc along x (columns), tc along y (rows)
    one = np.ones_like(sc)
    dirs = [np.stack([one, -tc, -sc], -1), np.stack([-one, -tc, sc], -1), np.stack([sc, one, tc], -1),
            np.stack([sc, -one, -tc], -1), np.stack([sc, -tc, one], -1), np.stack([-sc, -tc, -one], -1)]
    sun = np.asarray(sun_dir, np.float64)
    sun /= np.linalg.norm(sun)
    out = np.zeros((6, S, S, 4), np.float32)
    for f, d in enumerate(dirs):
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        up = np.clip(d[..., 1], 0, 1)
        sky = np.stack([0.25 + 0.2 * (1 - up), 0.45 + 0.25 * (1 - up), 0.9 - 0.1 * (1 - up)], -1)
        lobe = np.clip((d @ sun), 0, 1) ** 32
        col = sky + lobe[..., None] * np.array([4.0, 3.2, 2.0])
        ground = np.array([0.12, 0.10, 0.08])
        col = np.where((d[..., 1] < 0)[..., None], ground, col)
        out[f, ..., :3] = col * intensity
        out[f, ..., 3] = 1.0
    return out.astype(np.float16).view(np.uint16)


def prefiltered_chain(sky_u16: np.ndarray, levels: int = 5) -> np.ndarray:
    """Stand-in for dw::CubemapPrefiler (common.h:94): mip chain by 2x2 box filtering, levels packed back to
    back ([6][s][s][4] each, s = S >> level) as the hr_environment contract asks."""
    lvl = sky_u16.view(np.float16).astype(np.float32)
    out = []
    for _ in range(levels):
        out.append(lvl.astype(np.float16).reshape(-1))
        if lvl.shape[1] > 1:
            lvl = 0.25 * (lvl[:, 0::2, 0::2] + lvl[:, 1::2, 0::2] + lvl[:, 0::2, 1::2] + lvl[:, 1::2, 1::2])
    return np.ascontiguousarray(np.concatenate(out)).view(np.uint16)


def brdf_lut(size: int = 32) -> np.ndarray:
    """Stand-in for dw::BRDFIntegrateLUT (common.h:228): [size][size][2] fp16 (scale, bias) over (N.V, roughness),
    Karis' analytic fit of the split-sum integral."""
    nv = (np.arange(size, dtype=np.float64) + 0.5) / size
    r = (np.arange(size, dtype=np.float64) + 0.5) / size
    NV, R = np.meshgrid(nv, r)  # rows = roughness (v), cols = N.V (u)
    c0 = np.array([-1.0, -0.0275, -0.572, 0.022])
    c1 = np.array([1.0, 0.0425, 1.04, -0.04])
    rr = R[..., None] * c0 + c1
    a004 = np.minimum(rr[..., 0] * rr[..., 0], 2.0 ** (-9.28 * NV)) * rr[..., 0] + rr[..., 1]
    A = -1.04 * a004 + rr[..., 2]
    B = 1.04 * a004 + rr[..., 3]
    return np.ascontiguousarray(np.stack([A, B], -1).astype(np.float16)).view(np.uint16)


def sh9_from_cubemap(sky_u16: np.ndarray) -> np.ndarray:
    """Stand-in for dw::CubemapSHProjection (common.h:93): projects the sky cubemap onto 9 SH coefficients
    ([9][4] float32, rgb used) with the basis of deferred.frag:96-113."""
    sky = sky_u16.view(np.float16).astype(np.float64)[..., :3]
    S = sky.shape[1]
    t = (np.arange(S) + 0.5) / S * 2 - 1
    sc, tc = np.meshgrid(t, t)
    one = np.ones_like(sc)
    dirs = [np.stack([one, -tc, -sc], -1), np.stack([-one, -tc, sc], -1), np.stack([sc, one, tc], -1),
            np.stack([sc, -one, -tc], -1), np.stack([sc, -tc, one], -1), np.stack([-sc, -tc, -one], -1)]
    out = np.zeros((9, 4))
    wsum = 0.0
    for f, d in enumerate(dirs):
        r2 = (d ** 2).sum(-1)
        w = 4.0 / (r2 ** 1.5) / (S * S)                      # texel solid angle
        d = d / np.sqrt(r2)[..., None]
        x, y, z = d[..., 0], d[..., 1], d[..., 2]
        basis = [0.282095 * one, -0.488603 * y, 0.488603 * z, -0.488603 * x, 1.092548 * x * y, -1.092548 * y * z,
                 0.315392 * (3 * z * z - 1), -1.092548 * x * z, 0.546274 * (x * x - y * y)]
        for k, b in enumerate(basis):
            out[k, :3] += (sky[f] * (b * w)[..., None]).sum((0, 1))
        wsum += w.sum()
    out[:, :3] *= 4 * math.pi / wsum
    return out.astype(np.float32)
