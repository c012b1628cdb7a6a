"""Oracle wrappers for the reflections pass (TEST INFRASTRUCTURE ONLY; see pyoracle.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .pyoracle import _p, _ubo_ptr, c_f32p, c_u8p, c_u16p, lib, upsample
from .pyoracle_ddgi import _ddgi_ptr


class TraceParams(C.Structure):
    _fields_ = [("bias", C.c_float), ("trim", C.c_float), ("num_frames", C.c_uint32), ("sample_gi", C.c_int), ("approximate_with_ddgi", C.c_int),
                ("gi_intensity", C.c_float), ("rough_ddgi_intensity", C.c_float), ("ibl_indirect_specular_intensity", C.c_float)]


def ray_trace(scene, ubo, ddgi, cur, sobol, sr, prm: TraceParams, env, irr, dep):
    h, w = cur["depth"].shape
    out = np.zeros((h, w, 4), np.uint16)
    rays = C.c_uint64(0)
    lib().orc_reflections_ray_trace(scene.h, _ubo_ptr(ubo), _ddgi_ptr(ddgi), C.c_int(w), C.c_int(h), _p(cur["depth"], c_f32p), _p(cur["gb2"], c_u16p),
                                    _p(cur["gb3"], c_u16p), _p(sobol, c_u8p), _p(sr, c_u8p), C.byref(prm), _p(env["sky"], c_u16p),
                                    C.c_int(env["sky"].shape[1]), _p(env["prefiltered"], c_u16p), C.c_int(env["pre_size"]), C.c_int(env["pre_levels"]),
                                    _p(env["lut"], c_u16p), C.c_int(env["lut"].shape[0]), _p(irr, c_u16p), _p(dep, c_u16p), _p(out, c_u16p), C.byref(rays))
    return out, rays.value


def temporal(ubo, inp, cur, prev, hist_color, hist_moments, camera_delta, alpha, moments_alpha, approx):
    h, w = cur["depth"].shape
    oc, om = np.zeros((h, w, 4), np.uint16), np.zeros((h, w, 4), np.uint16)
    tiles = np.zeros(((h + 7) // 8, (w + 7) // 8), np.uint8)
    cd = np.ascontiguousarray(camera_delta, np.float32)
    lib().orc_reflections_temporal(_ubo_ptr(ubo), C.c_int(w), C.c_int(h), _p(inp, c_u16p), _p(cur["depth"], c_f32p), _p(cur["gb2"], c_u16p),
                                   _p(cur["gb3"], c_u16p), _p(prev["depth"], c_f32p), _p(prev["gb2"], c_u16p), _p(prev["gb3"], c_u16p),
                                   _p(hist_color, c_u16p), _p(hist_moments, c_u16p), _p(cd, c_f32p), C.c_float(alpha), C.c_float(moments_alpha),
                                   C.c_int(int(approx)), _p(oc, c_u16p), _p(om, c_u16p), _p(tiles, c_u8p))
    return oc, om, tiles


def atrous(inp, cur, tiles, step, radius, phi_color, phi_normal, sigma_depth, approx):
    h, w = inp.shape[:2]
    out = np.zeros((h, w, 4), np.uint16)
    lib().orc_reflections_atrous(C.c_int(w), C.c_int(h), _p(inp, c_u16p), _p(cur["depth"], c_f32p), _p(cur["gb2"], c_u16p), _p(cur["gb3"], c_u16p),
                                 _p(tiles, c_u8p), C.c_int(radius), C.c_int(step), C.c_float(phi_color), C.c_float(phi_normal), C.c_float(sigma_depth),
                                 C.c_int(int(approx)), _p(out, c_u16p))
    return out


class ReflectionsPass:
    """Host-side sequencing of RayTracedReflections::render (ray_traced_reflections.cpp:107-123) on the oracle."""

    def __init__(self, w, h, sample_gi=True, approximate_with_ddgi=True, gi_intensity=0.5, rough_ddgi_intensity=0.5,
                 ibl_indirect_specular_intensity=0.05, bias=0.5, trim=0.8, alpha=0.01, moments_alpha=0.2, blur_as_input=False, phi_color=10.0,
                 phi_normal=32.0, sigma_depth=1.0, radius=1, filter_iterations=4, feedback_iteration=1):
        self.w, self.h = w, h
        self.p = dict(sample_gi=sample_gi, approximate_with_ddgi=approximate_with_ddgi, gi_intensity=gi_intensity,
                      rough_ddgi_intensity=rough_ddgi_intensity, ibl_indirect_specular_intensity=ibl_indirect_specular_intensity, bias=bias,
                      trim=trim, alpha=alpha, moments_alpha=moments_alpha, blur_as_input=blur_as_input, phi_color=phi_color,
                      phi_normal=phi_normal, sigma_depth=sigma_depth, radius=radius, filter_iterations=filter_iterations,
                      feedback_iteration=feedback_iteration)
        z = lambda: np.zeros((h, w, 4), np.uint16)
        self.color = [z(), z()]      # current_output_image[2]
        self.moments = [z(), z()]
        self.prev_image = z()
        self.ping_pong = False
        self.stages = {}

    def render(self, scene, ubo, ddgi, cur, prev, sobol, sr, num_frames, env, irr, dep, camera_delta=(0, 0, 0), full=None, ping_pong=None):
        p = self.p
        pp = int(self.ping_pong if ping_pong is None else ping_pong)
        tp = TraceParams(p["bias"], p["trim"], num_frames, int(p["sample_gi"]), int(p["approximate_with_ddgi"]), p["gi_intensity"],
                         p["rough_ddgi_intensity"], p["ibl_indirect_specular_intensity"])
        traced, nrays = ray_trace(scene, ubo, ddgi, cur, sobol, sr, tp, env, irr, dep)
        hist = self.prev_image if p["blur_as_input"] else self.color[1 - pp]
        oc, om, tiles = temporal(ubo, traced, cur, prev, hist, self.moments[1 - pp], camera_delta, p["alpha"], p["moments_alpha"], p["approximate_with_ddgi"])
        self.color[pp], self.moments[pp] = oc, om
        img, its = oc, []
        for i in range(p["filter_iterations"]):
            img = atrous(img, cur, tiles, 1 << i, p["radius"], p["phi_color"], p["phi_normal"], p["sigma_depth"], p["approximate_with_ddgi"])
            its.append(img)
            if i == p["feedback_iteration"] and p["blur_as_input"]:
                self.prev_image = img.copy()
        up = upsample(full, cur, img, channels=4, sky_value=0.0, power=0.0) if full is not None else None
        self.stages = dict(trace=traced, rays=nrays, temporal=oc, moments=om, tiles=tiles, atrous=its, upsample=up, output=img if up is None else up)
        if ping_pong is None:
            self.ping_pong = not self.ping_pong
        return self.stages["output"]
