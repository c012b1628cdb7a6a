"""Python mirrors of GroundTruthPathTracer (src/ground_truth_path_tracer.h) and TemporalAA (src/temporal_aa.h)
over the C ABI — the SURVEY.md §8f rows 3 and 4."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api
from .api import _check, _stream_ptr, hr_band, hr_gbuffer_level, hr_image_view, hr_ubo, lib, view_to_tensor
from .api_deferred import _view


class hr_ground_truth_params(C.Structure):
    _fields_ = [("max_ray_bounces", C.c_int32), ("roughness_multiplier", C.c_float), ("trace_indirect", C.c_int32)]


class hr_taa_params(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("sharpen", C.c_int32), ("reset", C.c_int32), ("feedback_min", C.c_float), ("feedback_max", C.c_float)]


class GroundTruthPathTracer:
    """``render(scene, ubo, env)`` = GroundTruthPathTracer::render(cmd_buf); ``restart_accumulation()`` as upstream."""

    def __init__(self, ctx, width, height, band=None):
        self.h = C.c_void_p()
        self.params = hr_ground_truth_params()
        lib().hr_ground_truth_default_params(C.byref(self.params))
        b = hr_band(*band) if band else None
        _check(lib().hr_ground_truth_create(ctx.h, C.c_int32(width), C.c_int32(height), C.byref(b) if b else None, C.byref(self.h)), "hr_ground_truth_create")
        self.width, self.height = width, height

    def render(self, scene, np_ubo, env, stream=None):
        u = np_ubo if isinstance(np_ubo, hr_ubo) else api.make_ubo(np_ubo)
        _check(lib().hr_ground_truth_render(self.h, scene.h, C.byref(u), C.byref(env), C.byref(self.params), _stream_ptr(stream)), "hr_ground_truth_render")

    def output(self):
        v = hr_image_view()
        _check(lib().hr_ground_truth_output(self.h, C.byref(v)), "hr_ground_truth_output")
        return view_to_tensor(v)

    def restart_accumulation(self):
        _check(lib().hr_ground_truth_restart_accumulation(self.h), "hr_ground_truth_restart_accumulation")

    def ray_count(self) -> int:
        n = C.c_uint64(0)
        _check(lib().hr_ground_truth_ray_count(self.h, C.byref(n)), "hr_ground_truth_ray_count")
        return n.value

    def close(self):
        if self.h:
            lib().hr_ground_truth_destroy(self.h)
            self.h = C.c_void_p()


class TemporalAA:
    """``update(num_frames)`` = TemporalAA::update (returns (current.xy, prev.xy) jitter); ``render(color, gbuffer,
    ping_pong)`` = TemporalAA::render; ``output(ping_pong)`` = output_ds."""

    def __init__(self, ctx, width, height):
        self.h = C.c_void_p()
        self.params = hr_taa_params()
        lib().hr_taa_default_params(C.byref(self.params))
        _check(lib().hr_taa_create(ctx.h, C.c_int32(width), C.c_int32(height), C.byref(self.h)), "hr_taa_create")
        self.width, self.height = width, height

    def update(self, num_frames):
        j = (C.c_float * 4)()
        _check(lib().hr_taa_update(self.h, C.c_uint32(num_frames), C.byref(self.params), j), "hr_taa_update")
        return np.array(list(j), np.float32)

    def render(self, color, gb, ping_pong, stream=None):
        """color: cuda fp16 [H,W,4]; gb: dict of cuda tensors (gb2, depth, ...) at full resolution"""
        lvl = api.gbuffer_level(gb)
        _check(lib().hr_taa_render(self.h, C.byref(_view(color)), C.byref(lvl), C.c_int32(int(ping_pong)), C.byref(self.params), _stream_ptr(stream)), "hr_taa_render")

    def output(self, ping_pong):
        v = hr_image_view()
        _check(lib().hr_taa_output(self.h, C.c_int32(int(ping_pong)), C.byref(v)), "hr_taa_output")
        return view_to_tensor(v)

    def close(self):
        if self.h:
            lib().hr_taa_destroy(self.h)
            self.h = C.c_void_p()


def tone_map(ctx, color, single_channel=False, exposure=1.0, want_rgba8=True, stream=None):
    """ToneMap::render (tone_map.cpp:98-143): color cuda fp16 [H,W,4] -> (fp32 [H,W,4], uint8 [H,W,4] or None)"""
    import torch
    h, w = color.shape[:2]
    out_f = torch.empty((h, w, 4), dtype=torch.float32, device=color.device)
    out_b = torch.empty((h, w, 4), dtype=torch.uint8, device=color.device) if want_rgba8 else None
    _check(lib().hr_tone_map(ctx.h, C.byref(_view(color)), C.c_int32(int(single_channel)), C.c_float(exposure), api._ptr(out_f), api._ptr(out_b),
                             _stream_ptr(stream)), "hr_tone_map")
    return out_f, out_b


api.ABI_SYMBOLS += ["hr_tone_map", "hr_ground_truth_default_params", "hr_ground_truth_create", "hr_ground_truth_render", "hr_ground_truth_output",
                    "hr_ground_truth_restart_accumulation", "hr_ground_truth_ray_count", "hr_ground_truth_set_profiling",
                    "hr_ground_truth_get_stage_times", "hr_ground_truth_destroy",
                    "hr_taa_default_params", "hr_taa_create", "hr_taa_update", "hr_taa_render", "hr_taa_output", "hr_taa_set_profiling",
                    "hr_taa_get_stage_times", "hr_taa_destroy"]
