"""Oracle wrappers for the DDGI pass (TEST INFRASTRUCTURE ONLY; see pyoracle.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .pyoracle import _p, _ubo_ptr, c_f32p, c_u16p, lib


def _ddgi_ptr(u: np.ndarray):
    assert u.nbytes == 88
    return C.c_void_p(u.ctypes.data)


def ray_trace(scene, ubo, ddgi, orientation, num_frames, infinite_bounces, gi_intensity, sky, prev_irr, prev_depth):
    n_probes = int(np.prod(ddgi["probe_counts"]))
    R = int(ddgi["rays_per_probe"])
    rad = np.zeros((n_probes, R, 4), np.uint16)
    dd = np.zeros((n_probes, R, 4), np.uint16)
    rays = C.c_uint64(0)
    orientation = np.ascontiguousarray(orientation, np.float32)
    lib().orc_ddgi_ray_trace(scene.h, _ubo_ptr(ubo), _ddgi_ptr(ddgi), _p(orientation, c_f32p), C.c_uint32(num_frames), C.c_int(int(infinite_bounces)),
                             C.c_float(gi_intensity), _p(sky, c_u16p), C.c_int(sky.shape[1]), _p(prev_irr, c_u16p), _p(prev_depth, c_u16p),
                             _p(rad, c_u16p), _p(dd, c_u16p), C.byref(rays))
    return rad, dd, rays.value


def probe_update(ddgi, depth_probe, first_frame, rad, dd, prev_atlas):
    out = prev_atlas.copy() * 0
    lib().orc_ddgi_probe_update(_ddgi_ptr(ddgi), C.c_int(int(depth_probe)), C.c_int(int(first_frame)), _p(rad, c_u16p), _p(dd, c_u16p),
                                _p(prev_atlas, c_u16p), _p(out, c_u16p))
    return out


def border_update(ddgi, depth_probe, atlas):
    lib().orc_ddgi_border_update(_ddgi_ptr(ddgi), C.c_int(int(depth_probe)), _p(atlas, c_u16p))
    return atlas


def sample_probe_grid(ubo, ddgi, depth, gb2, gi_intensity, irr, dep):
    h, w = depth.shape
    out = np.zeros((h, w, 4), np.uint16)
    lib().orc_ddgi_sample_probe_grid(_ubo_ptr(ubo), _ddgi_ptr(ddgi), C.c_int(w), C.c_int(h), _p(depth, c_f32p), _p(gb2, c_u16p), C.c_float(gi_intensity),
                                     _p(irr, c_u16p), _p(dep, c_u16p), _p(out, c_u16p))
    return out


class DDGIPass:
    """Host-side sequencing of DDGI::render (ddgi.cpp:89-104) on the oracle."""

    def __init__(self, ddgi: np.ndarray, infinite_bounces=True, infinite_bounce_intensity=1.7, gi_intensity=1.0):
        self.ddgi = ddgi
        self.p = dict(infinite_bounces=infinite_bounces, infinite_bounce_intensity=infinite_bounce_intensity, gi_intensity=gi_intensity)
        iw, ih = int(ddgi["irradiance_texture_width"]), int(ddgi["irradiance_texture_height"])
        dw, dh = int(ddgi["depth_texture_width"]), int(ddgi["depth_texture_height"])
        self.irr = [np.zeros((ih, iw, 4), np.uint16) for _ in range(2)]
        self.dep = [np.zeros((dh, dw, 2), np.uint16) for _ in range(2)]
        self.first_frame, self.ping_pong = True, False
        self.stages = {}

    def current_read(self):
        """DDGI::current_read_ds(): the atlases written by the last render()."""
        i = int(not self.ping_pong)
        return self.irr[i], self.dep[i]

    def render(self, scene, ubo, cur, sky, orientation, num_frames):
        p, d = self.p, self.ddgi
        rd, wr = int(not self.ping_pong), int(self.ping_pong)
        inf = p["infinite_bounces"] and not self.first_frame   # ddgi.cpp:790
        rad, dd, rays = ray_trace(scene, ubo, d, orientation, num_frames, inf, p["infinite_bounce_intensity"], sky, self.irr[rd], self.dep[rd])
        self.irr[wr] = border_update(d, False, probe_update(d, False, self.first_frame, rad, dd, self.irr[rd]))
        self.dep[wr] = border_update(d, True, probe_update(d, True, self.first_frame, rad, dd, self.dep[rd]))
        out = sample_probe_grid(ubo, d, cur["depth"], cur["gb2"], p["gi_intensity"], self.irr[wr], self.dep[wr])
        self.stages = dict(radiance=rad, direction_distance=dd, rays=rays, irradiance=self.irr[wr], depth=self.dep[wr], output=out)
        self.first_frame = False
        self.ping_pong = not self.ping_pong
        return out
