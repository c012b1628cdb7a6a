"""Python mirrors of DDGI (src/ddgi.h) and the environment inputs, over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api
from .api import _Pass, _check, _ptr, _stream_ptr, hr_image_view, lib, view_to_tensor


class hr_environment(C.Structure):
    _fields_ = [("sky", C.c_void_p), ("sky_size", C.c_int32), ("prefiltered", C.c_void_p), ("prefiltered_size", C.c_int32),
                ("prefiltered_levels", C.c_int32), ("brdf_lut", C.c_void_p), ("brdf_lut_size", C.c_int32)]


class hr_ddgi_uniforms(C.Structure):
    _fields_ = [("grid_start_position", C.c_float * 3), ("grid_step", C.c_float * 3), ("probe_counts", C.c_int32 * 3),
                ("max_distance", C.c_float), ("depth_sharpness", C.c_float), ("hysteresis", C.c_float), ("normal_bias", C.c_float),
                ("energy_preservation", C.c_float), ("irradiance_probe_side_length", C.c_int32), ("irradiance_texture_width", C.c_int32),
                ("irradiance_texture_height", C.c_int32), ("depth_probe_side_length", C.c_int32), ("depth_texture_width", C.c_int32),
                ("depth_texture_height", C.c_int32), ("rays_per_probe", C.c_int32), ("visibility_test", C.c_int32)]


assert C.sizeof(hr_ddgi_uniforms) == 88


class hr_ddgi_params(C.Structure):
    _fields_ = [("infinite_bounces", C.c_int32), ("infinite_bounce_intensity", C.c_float), ("gi_intensity", C.c_float),
                ("random_orientation", C.c_float * 9), ("exact", C.c_int32)]


def environment(sky, prefiltered=None, prefiltered_size=0, prefiltered_levels=0, brdf_lut=None) -> hr_environment:
    """sky: cuda fp16 tensor [6,S,S,4]."""
    e = hr_environment()
    e.sky, e.sky_size = _ptr(sky), int(sky.shape[1])
    e.prefiltered, e.prefiltered_size, e.prefiltered_levels = _ptr(prefiltered), int(prefiltered_size), int(prefiltered_levels)
    e.brdf_lut, e.brdf_lut_size = _ptr(brdf_lut), int(brdf_lut.shape[0]) if brdf_lut is not None else 0
    e._keep = (sky, prefiltered, brdf_lut)
    return e


def grid_from_extents(lo, hi, probe_distance: float, rays_per_probe: int = 256) -> np.ndarray:
    """hr_ddgi_grid_from_extents = DDGI::initialize_probe_grid + atlas sizing + uniform constants (ddgi.cpp:150-169, :197-201, :738-763); host only.
    Returns the 88-byte block as a numpy record (synth_env.DDGI_DTYPE)."""
    from . import synth_env
    u = hr_ddgi_uniforms()
    a, b = (C.c_float * 3)(*[float(v) for v in lo]), (C.c_float * 3)(*[float(v) for v in hi])
    _check(lib().hr_ddgi_grid_from_extents(a, b, C.c_float(probe_distance), C.c_int32(rays_per_probe), C.byref(u)), "hr_ddgi_grid_from_extents")
    out = np.zeros((), synth_env.DDGI_DTYPE)
    C.memmove(out.ctypes.data, C.byref(u), 88)
    return out


def make_uniforms(np_ddgi: np.ndarray) -> hr_ddgi_uniforms:
    assert np_ddgi.nbytes == 88
    u = hr_ddgi_uniforms()
    C.memmove(C.byref(u), np_ddgi.ctypes.data, 88)
    return u


class DDGI(_Pass):
    """src/ddgi.h:6-135.  ``render(scene, inputs, env, orientation)`` = DDGI::render(cmd_buf)."""
    _prefix = "hr_ddgi"
    IMG_RADIANCE, IMG_DIRDIST, IMG_IRR0, IMG_IRR1, IMG_DEPTH0, IMG_DEPTH1, IMG_SAMPLE = range(7)

    def __init__(self, ctx, width, height, np_ddgi, scale=api.SCALE_FULL_RES):
        self.ctx = ctx
        self.uniforms = make_uniforms(np_ddgi)
        self.params = hr_ddgi_params()
        lib().hr_ddgi_default_params(C.byref(self.params))
        self.h = C.c_void_p()
        _check(lib().hr_ddgi_create(ctx.h, C.c_int32(width), C.c_int32(height), C.c_int(scale), C.byref(self.uniforms), C.byref(self.h)), "hr_ddgi_create")
        self.width, self.height = width >> scale, height >> scale

    def set_normal_bias(self, v: float):
        """DDGI::set_normal_bias (ddgi.h:29)"""
        _check(lib().hr_ddgi_set_normal_bias(self.h, C.c_float(v)), "hr_ddgi_set_normal_bias")
        self.uniforms.normal_bias = float(v)

    def set_orientation(self, m9):
        for i in range(9):
            self.params.random_orientation[i] = float(m9[i])

    def render(self, scene, inputs, env, orientation=None, stream=None):
        if orientation is not None:
            self.set_orientation(orientation)
        _check(lib().hr_ddgi_render(self.h, scene.h, C.byref(inputs), C.byref(env), C.byref(self.params), _stream_ptr(stream)), "hr_ddgi_render")

    def output(self, kind=None):
        v = hr_image_view()
        _check(lib().hr_ddgi_output(self.h, C.byref(v)), "hr_ddgi_output")
        return view_to_tensor(v)

    # ---- stage-by-stage (ddgi.cpp:89-104) and multi-GPU sharding
    def ray_trace(self, scene, inputs, env, stream=None):
        _check(lib().hr_ddgi_ray_trace(self.h, scene.h, C.byref(inputs), C.byref(env), C.byref(self.params), _stream_ptr(stream)), "hr_ddgi_ray_trace")

    def trace_stats(self, scene, inputs, env, stream=None):
        """(rays, BVH node steps, triangle tests) of the ray-trace stage from the instrumented kernel (probe rays + light / sky rays)."""
        out = (C.c_uint64 * 3)()
        _check(lib().hr_ddgi_trace_stats(self.h, scene.h, C.byref(inputs), C.byref(env), C.byref(self.params), out, _stream_ptr(stream)), "hr_ddgi_trace_stats")
        return int(out[0]), int(out[1]), int(out[2])

    def probe_update(self, stream=None):
        _check(lib().hr_ddgi_probe_update(self.h, _stream_ptr(stream)), "hr_ddgi_probe_update")

    def sample_probe_grid(self, inputs, stream=None):
        _check(lib().hr_ddgi_sample_probe_grid(self.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_ddgi_sample_probe_grid")

    def end_frame(self):
        _check(lib().hr_ddgi_end_frame(self.h), "hr_ddgi_end_frame")

    def set_shard(self, probe_z0, probe_z1, row_y0, row_y1):
        _check(lib().hr_ddgi_set_shard(self.h, C.c_int32(probe_z0), C.c_int32(probe_z1), C.c_int32(row_y0), C.c_int32(row_y1)), "hr_ddgi_set_shard")

    def current_write(self):
        a, b = hr_image_view(), hr_image_view()
        _check(lib().hr_ddgi_current_write(self.h, C.byref(a), C.byref(b)), "hr_ddgi_current_write")
        return view_to_tensor(a), view_to_tensor(b)

    def current_read(self):
        a, b = hr_image_view(), hr_image_view()
        _check(lib().hr_ddgi_current_read(self.h, C.byref(a), C.byref(b)), "hr_ddgi_current_read")
        return view_to_tensor(a), view_to_tensor(b)

    def restart_accumulation(self):
        _check(lib().hr_ddgi_restart_accumulation(self.h), "hr_ddgi_restart_accumulation")

    def ray_count(self) -> int:
        n = C.c_uint64(0)
        _check(lib().hr_ddgi_ray_count(self.h, C.byref(n)), "hr_ddgi_ray_count")
        return n.value


api.ABI_SYMBOLS += ["hr_ddgi_default_params", "hr_ddgi_create", "hr_ddgi_render", "hr_ddgi_output", "hr_ddgi_current_read",
                    "hr_ddgi_restart_accumulation", "hr_ddgi_destroy", "hr_ddgi_ray_trace", "hr_ddgi_probe_update", "hr_ddgi_sample_probe_grid",
                    "hr_ddgi_end_frame", "hr_ddgi_image", "hr_ddgi_get_uniforms", "hr_ddgi_set_profiling", "hr_ddgi_get_stage_times", "hr_ddgi_ray_count", "hr_ddgi_set_shard", "hr_ddgi_current_write", "hr_ddgi_trace_stats",
                    "hr_ddgi_grid_from_extents", "hr_ddgi_set_normal_bias"]
