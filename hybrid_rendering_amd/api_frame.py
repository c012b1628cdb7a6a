"""Python mirror of hr_hybrid_frame (include/hr_api.h): the four passes of a frame enqueued as the dependency graph they form."""
from __future__ import annotations

import ctypes as C

from . import api
from .api import _check, _stream_ptr, lib

FRAME_SERIAL, FRAME_STREAMS, FRAME_GRAPH = 0, 1, 2


class hr_hybrid_frame_desc(C.Structure):
    _fields_ = [("environment", C.c_void_p), ("shadows_inputs", C.c_void_p), ("shadows_params", C.c_void_p), ("ao_inputs", C.c_void_p), ("ao_params", C.c_void_p),
                ("ddgi_inputs", C.c_void_p), ("ddgi_params", C.c_void_p), ("reflections_inputs", C.c_void_p), ("reflections_params", C.c_void_p)]


def _addr(obj):
    return C.cast(C.pointer(obj), C.c_void_p) if obj is not None else None


class HybridFrame:
    """hr_hybrid_frame over existing pass objects (not owned): render(scene, env, inputs..., mode) = the reference's four render() calls
    of main.cpp:80-83, serial / forked over streams / as one hipGraph."""

    def __init__(self, ctx, shadows=None, ao=None, ddgi=None, reflections=None):
        self.h = C.c_void_p()
        self.passes = (shadows, ao, ddgi, reflections)
        hs = [p.h if p is not None else None for p in self.passes]
        _check(lib().hr_hybrid_frame_create(ctx.h, hs[0], hs[1], hs[2], hs[3], C.byref(self.h)), "hr_hybrid_frame_create")

    def render(self, scene, env, shadows_inputs=None, ao_inputs=None, ddgi_inputs=None, reflections_inputs=None, mode=FRAME_STREAMS, stream=None):
        sh, ao, gi, rf = self.passes
        d = hr_hybrid_frame_desc()
        d.environment = _addr(env)
        if sh is not None:
            d.shadows_inputs, d.shadows_params = _addr(shadows_inputs), _addr(sh.params)
        if ao is not None:
            d.ao_inputs, d.ao_params = _addr(ao_inputs), _addr(ao.params)
        if gi is not None:
            d.ddgi_inputs, d.ddgi_params = _addr(ddgi_inputs), _addr(gi.params)
        if rf is not None:
            d.reflections_inputs, d.reflections_params = _addr(reflections_inputs), _addr(rf.params)
        _check(lib().hr_hybrid_frame_render(self.h, scene.h, C.byref(d), C.c_int32(mode), _stream_ptr(stream)), "hr_hybrid_frame_render")

    def graph_stats(self):
        a, b = C.c_int32(0), C.c_int32(0)
        _check(lib().hr_hybrid_frame_graph_stats(self.h, C.byref(a), C.byref(b)), "hr_hybrid_frame_graph_stats")
        return a.value, b.value

    def close(self):
        if self.h:
            lib().hr_hybrid_frame_destroy(self.h)
            self.h = C.c_void_p()


api.ABI_SYMBOLS += ["hr_hybrid_frame_create", "hr_hybrid_frame_render", "hr_hybrid_frame_graph_stats", "hr_hybrid_frame_destroy", "hr_hybrid_frame_fork",
                    "hr_hybrid_frame_join"]
