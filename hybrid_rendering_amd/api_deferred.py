"""Python mirror of the DeferredShading composite (src/deferred_shading.h) over the C ABI."""
from __future__ import annotations

import ctypes as C

from . import api
from .api import _check, _stream_ptr, hr_image_view, lib, view_to_tensor


class hr_deferred_params(C.Structure):
    _fields_ = [("use_ray_traced_shadows", C.c_int32), ("use_ray_traced_ao", C.c_int32), ("use_ray_traced_reflections", C.c_int32),
                ("use_ddgi", C.c_int32), ("irradiance_sh9", (C.c_float * 4) * 9), ("draw_skybox", C.c_int32)]


def _view(t):
    """cuda fp16 tensor [H,W] / [H,W,2] / [H,W,4] -> hr_image_view."""
    if t is None:
        return None
    ch = 1 if t.dim() == 2 else int(t.shape[2])
    fmt = {1: 2, 2: 3, 4: 4}[ch]  # HR_FORMAT_R16F / RG16F / RGBA16F
    return hr_image_view(C.c_void_p(t.data_ptr()), int(t.shape[1]), int(t.shape[0]), int(t.shape[1]) * 2 * ch, fmt)


class DeferredShading:
    """src/deferred_shading.h:15-60.  ``render(inputs, env, shadow, ao, reflections, gi)`` = DeferredShading::render."""

    def __init__(self, ctx, width, height):
        self.h = C.c_void_p()
        self.params = hr_deferred_params()
        lib().hr_deferred_default_params(C.byref(self.params))
        _check(lib().hr_deferred_create(ctx.h, C.c_int32(width), C.c_int32(height), C.byref(self.h)), "hr_deferred_create")

    def set_sh9(self, sh9):
        for k in range(9):
            for c in range(4):
                self.params.irradiance_sh9[k][c] = float(sh9[k][c])

    def render(self, inputs, env, shadow=None, ao=None, reflections=None, gi=None, stream=None):
        vs = [_view(t) for t in (shadow, ao, reflections, gi)]
        ptr = lambda v: C.byref(v) if v is not None else None
        _check(lib().hr_deferred_render(self.h, C.byref(inputs), C.byref(env), ptr(vs[0]), ptr(vs[1]), ptr(vs[2]), ptr(vs[3]), C.byref(self.params),
                                        _stream_ptr(stream)), "hr_deferred_render")

    def output(self):
        v = hr_image_view()
        _check(lib().hr_deferred_output(self.h, C.byref(v)), "hr_deferred_output")
        return view_to_tensor(v)

    def close(self):
        if self.h:
            lib().hr_deferred_destroy(self.h)
            self.h = C.c_void_p()


api.ABI_SYMBOLS += ["hr_deferred_default_params", "hr_deferred_create", "hr_deferred_render", "hr_deferred_output", "hr_deferred_destroy"]
