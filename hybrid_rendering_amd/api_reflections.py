"""Python mirror of RayTracedReflections (src/ray_traced_reflections.h) over the C ABI."""
from __future__ import annotations

import ctypes as C

from . import api
from .api import _Pass, _check, _stream_ptr, lib


class hr_reflections_params(C.Structure):
    _fields_ = [("denoise", C.c_int32), ("sample_gi", C.c_int32), ("approximate_with_ddgi", C.c_int32), ("gi_intensity", C.c_float),
                ("rough_ddgi_intensity", C.c_float), ("ibl_indirect_specular_intensity", C.c_float), ("bias", C.c_float), ("trim", C.c_float),
                ("alpha", C.c_float), ("moments_alpha", C.c_float), ("blur_as_input", C.c_int32), ("phi_color", C.c_float),
                ("phi_normal", C.c_float), ("sigma_depth", C.c_float), ("radius", C.c_int32), ("filter_iterations", C.c_int32),
                ("feedback_iteration", C.c_int32), ("camera_delta", C.c_float * 3), ("frame_time", C.c_float), ("exact", C.c_int32)]


class RayTracedReflections(_Pass):
    """src/ray_traced_reflections.h:8-150.  ``render(scene, inputs, env, ddgi)`` = RayTracedReflections::render(cmd_buf, ddgi)."""
    _prefix = "hr_reflections"
    IMG_TRACE, IMG_COLOR0, IMG_COLOR1, IMG_MOMENTS0, IMG_MOMENTS1, IMG_PREV, IMG_ATROUS0, IMG_ATROUS1, IMG_UPSAMPLE, IMG_TILES = range(10)

    def __init__(self, ctx, width, height, scale=api.SCALE_HALF_RES, band=None):
        self.ctx = ctx
        self.params = hr_reflections_params()
        lib().hr_reflections_default_params(C.byref(self.params))
        self.h = C.c_void_p()
        b = api.hr_band(*band) if band else None
        _check(lib().hr_reflections_create(ctx.h, C.c_int32(width), C.c_int32(height), C.c_int(scale), C.byref(b) if b else None, C.byref(self.h)),
               "hr_reflections_create")
        self.scale = scale
        self.width, self.height = width >> scale, height >> scale

    def set_camera_delta(self, d):
        for i in range(3):
            self.params.camera_delta[i] = float(d[i])

    def render(self, scene, inputs, env, ddgi, stream=None):
        if ddgi is None or not getattr(ddgi, "h", None):
            raise api.HRError("RayTracedReflections.render needs the DDGI pass (ray_traced_reflections.h:27: render(cmd_buf, DDGI*))")
        _check(lib().hr_reflections_render(self.h, scene.h, C.byref(inputs), C.byref(env), ddgi.h, C.byref(self.params), _stream_ptr(stream)),
               "hr_reflections_render")

    def atrous_iteration(self, inputs, i, stream=None):
        """one iteration of a_trous_filter (ray_traced_reflections.cpp:1143-1256) on the pass's current images: iteration 0 reads the temporal colour image,
        iteration i > 0 the a-trous image iteration i - 1 wrote (IMG_ATROUS1 for even i - 1, IMG_ATROUS0 for odd); writes the other one"""
        _check(lib().hr_reflections_atrous_iteration(self.h, C.byref(inputs), C.byref(self.params), C.c_int32(i), _stream_ptr(stream)), "hr_reflections_atrous_iteration")

    def trace_stats(self, scene, inputs, env, ddgi, stream=None):
        """(rays, BVH node steps, triangle tests) of the ray-trace stage from the instrumented kernel (reflection rays + light rays)."""
        out = (C.c_uint64 * 3)()
        _check(lib().hr_reflections_trace_stats(self.h, scene.h, C.byref(inputs), C.byref(env), ddgi.h, C.byref(self.params), out, _stream_ptr(stream)),
               "hr_reflections_trace_stats")
        return int(out[0]), int(out[1]), int(out[2])

    def ray_count(self) -> int:
        n = C.c_uint64(0)
        _check(lib().hr_reflections_ray_count(self.h, C.byref(n)), "hr_reflections_ray_count")
        return n.value


api.ABI_SYMBOLS += ["hr_reflections_default_params", "hr_reflections_create", "hr_reflections_render", "hr_reflections_output",
                    "hr_reflections_reset_history", "hr_reflections_destroy", "hr_reflections_ray_trace", "hr_reflections_denoise", "hr_reflections_temporal",
                    "hr_reflections_atrous_iteration", "hr_reflections_upsample", "hr_reflections_image", "hr_reflections_history_apron_exceeded", "hr_reflections_set_profiling",
                    "hr_reflections_get_stage_times", "hr_reflections_ray_count", "hr_reflections_trace_stats"]
