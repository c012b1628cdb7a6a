"""Oracle wrapper for the deferred composite (TEST INFRASTRUCTURE ONLY; see pyoracle.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .pyoracle import _p, _ubo_ptr, c_f32p, c_u8p, c_u16p, lib


def shade(ubo, gb, shadow, ao, reflections, gi, flags, sh9, env, skybox=True):
    """shadow: uint16 [H,W,C] or None; ao: [H,W] / [H,W,1]; reflections, gi: [H,W,4].  flags: bit0 shadow, 1 ao, 2 reflections, 3 gi.
    skybox: DeferredShading::render_skybox afterwards (sky texels take env["sky"]); False = render_shading alone (deferred.frag)"""
    h, w = gb["depth"].shape
    out = np.zeros((h, w, 4), np.uint16)
    ch = lambda a: 1 if (a is None or a.ndim == 2) else a.shape[2]
    sh9 = np.ascontiguousarray(sh9, np.float32)
    lib().orc_deferred_shade(_ubo_ptr(ubo), C.c_int(w), C.c_int(h), _p(gb["gb1"], c_u8p), _p(gb["gb2"], c_u16p), _p(gb["gb3"], c_u16p),
                             _p(gb["depth"], c_f32p), _p(shadow, c_u16p), C.c_int(ch(shadow)), _p(ao, c_u16p), C.c_int(ch(ao)), _p(reflections, c_u16p),
                             _p(gi, c_u16p), C.c_int(flags), _p(sh9, c_f32p), _p(env["prefiltered"], c_u16p), C.c_int(env["pre_size"]),
                             C.c_int(env["pre_levels"]), _p(env["lut"], c_u16p), C.c_int(env["lut"].shape[0]), _p(out, c_u16p))
    if skybox and env.get("sky") is not None:
        lib().orc_deferred_skybox(_ubo_ptr(ubo), C.c_int(w), C.c_int(h), _p(gb["depth"], c_f32p), _p(env["sky"], c_u16p), C.c_int(env["sky"].shape[1]), _p(out, c_u16p))
    return out
