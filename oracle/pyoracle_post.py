"""Oracle wrappers for the ground-truth path tracer and TAA (TEST INFRASTRUCTURE ONLY; see pyoracle.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .pyoracle import _p, _ubo_ptr, c_f32p, c_u16p, lib


class GroundTruthPass:
    """Host sequencing of GroundTruthPathTracer::render (ground_truth_path_tracer.cpp:44-111) on the oracle."""

    def __init__(self, w, h, roughness_multiplier=1.0, band=None, max_ray_bounces=2, trace_indirect=False):
        self.w, self.h = w, h
        self.roughness_multiplier = roughness_multiplier
        # trace_indirect: the recursive traceRayEXT the reference ships commented out (rchit:95-105), re-enabled (extension)
        self.max_ray_bounces, self.trace_indirect = max_ray_bounces, trace_indirect
        self.y0, self.y1 = band if band else (0, h)
        self.images = [np.zeros((h, w, 4), np.uint16) for _ in range(2)]
        self.frame_idx, self.ping_pong = 0, False
        self.rays = 0

    def restart_accumulation(self):
        self.frame_idx = 0

    def render(self, scene, ubo, sky):
        if self.frame_idx == 0:
            self.ping_pong = False
        rd, wr = int(self.ping_pong), int(not self.ping_pong)
        rays = C.c_uint64(0)
        lib().orc_ground_truth_render_ex(scene.h, _ubo_ptr(ubo), _p(sky, c_u16p), C.c_int(sky.shape[1]), C.c_int(self.w), C.c_int(self.h),
                                         C.c_int(self.y0), C.c_int(self.y1), C.c_uint32(self.frame_idx), C.c_float(self.roughness_multiplier),
                                         C.c_int(self.max_ray_bounces), C.c_int(int(self.trace_indirect)),
                                         _p(self.images[rd], c_u16p), _p(self.images[wr], c_u16p), C.byref(rays))
        self.frame_idx += 1
        self.rays = rays.value
        self.ping_pong = not self.ping_pong
        return self.output()

    def output(self):
        return self.images[int(self.ping_pong)]


def halton(base, index):
    lib().orc_halton.restype = C.c_float
    return float(lib().orc_halton(C.c_int(base), C.c_int(index)))


class TAAPass:
    """Host sequencing of TemporalAA::update / render (temporal_aa.cpp:64-172) on the oracle."""

    def __init__(self, w, h, enabled=True, sharpen=True, reset=True, feedback_min=0.88, feedback_max=0.97):
        self.w, self.h = w, h
        self.enabled, self.sharpen, self.reset = enabled, sharpen, reset
        self.feedback_min, self.feedback_max = feedback_min, feedback_max
        self.images = [np.zeros((h, w, 4), np.uint16) for _ in range(2)]
        self.jitter = np.zeros(4, np.float32)

    def update(self, num_frames):
        prev_current = np.ascontiguousarray(self.jitter[:2].copy())
        j = np.zeros(4, np.float32)
        lib().orc_taa_jitter(C.c_uint32(num_frames), C.c_int(self.w), C.c_int(self.h), C.c_int(int(self.enabled)), _p(prev_current, c_f32p), _p(j, c_f32p))
        self.jitter = j
        return j

    def render(self, color, gb, ping_pong):
        if not self.enabled:
            return
        wr, rd = int(bool(ping_pong)), int(not ping_pong)
        if self.reset:
            self.images[rd][...] = color
        lib().orc_taa_resolve(C.c_int(self.w), C.c_int(self.h), _p(color, c_u16p), _p(self.images[rd], c_u16p), _p(gb["gb2"], c_u16p), _p(gb["depth"], c_f32p),
                              _p(self.jitter, c_f32p), C.c_float(self.feedback_min), C.c_float(self.feedback_max), C.c_int(int(self.sharpen)), _p(self.images[wr], c_u16p))

    def output(self, ping_pong):
        return self.images[int(bool(ping_pong))]


def tone_map(color, single_channel=False, exposure=1.0):
    """ToneMap::render (tone_map.cpp:98-143, tone_map.frag:50-68): color [h][w][4] fp16 bits -> FS_OUT_Color [h][w][4] fp32"""
    h, w = color.shape[:2]
    out = np.zeros((h, w, 4), np.float32)
    lib().orc_tone_map(C.c_int(w), C.c_int(h), _p(color, c_u16p), C.c_int(int(single_channel)), C.c_float(exposure), _p(out, c_f32p))
    return out
